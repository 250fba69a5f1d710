"""The small-M products of the views' chain under every (kernel variant, K slices) the planner could pick, against its own
choice.  python tools/exp/smallm_plans.py"""
import os
import torch
from od_wscl_amd import gemm, precision

precision.set_precision("bf16")
dev = torch.device("cuda:0")


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (M, N, K, what) in ((300, 4096, 12288, "views fc7 fwd, one label"), (896, 4096, 12288, "views fc7 fwd, three labels"),
                        (300, 4096, 4096, "views fc7 dgrad"), (896, 4096, 4096, "views fc7 dgrad, three labels"),
                        (300, 128, 12288, "views sim2 fwd"), (2000, 4096, 12288, "clean sim0 fwd")):
    a = (torch.randn(M, K, device=dev) * 0.1).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, device=dev)
    os.environ.pop("ODW_GEMM_VARIANT", None)
    os.environ.pop("ODW_GEMM_SPLITK", None)
    gemm._PLAN_CACHE.clear()
    base = timed(lambda: gemm.gemm_nt(a, b, M, N, K, out))
    res = []
    for var in ("ring", "big", "glds"):
        for sp in (1, 2, 3, 4, 6, 8, 12, 16):
            if var == "glds" and sp > 1:
                continue
            os.environ["ODW_GEMM_VARIANT"] = var
            os.environ["ODW_GEMM_SPLITK"] = str(sp)
            gemm._PLAN_CACHE.clear()
            try:
                t = timed(lambda: gemm.gemm_nt(a, b, M, N, K, out), reps=15)
            except Exception as e:          # (a split the planner refuses)
                continue
            res.append((t, var, sp))
    res.sort()
    print("%-34s M=%d N=%d K=%d: planner %.1f us; best forced: %s" % (what, M, N, K, base,
          ", ".join("%s x%d %.1f" % (v, s, t) for t, v, s in res[:5])))
