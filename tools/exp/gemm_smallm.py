"""Kernel variants on the sampled-view shapes of the step (M ~ 450-1100 rows): the planner's pick vs each forced variant."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from od_wscl_amd import gemm
from gemm_bench import timeit  # noqa
shapes = []
for M in (300, 450, 584, 746, 796, 858, 914, 1092, 1300, 1600):
    shapes.append(("fc6_dgrad", M, 25088, 4096))
for M in (584, 796, 914, 1092):
    shapes.append(("fc7_dgrad", M, 4096, 4096))
    shapes.append(("fc7_fwd3", M, 4096, 12288))
for name, M, N, K in shapes:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    res = {"shape": "%s %dx%dx%d" % (name, M, N, K)}
    for var in ("plan", "glds", "ring", "big"):
        if var == "plan":
            os.environ.pop("ODW_GEMM_VARIANT", None)
        else:
            os.environ["ODW_GEMM_VARIANT"] = var
        ms = timeit(lambda: gemm.gemm_nt(a, b, M, N, K, out), iters=30)
        res[var] = round(ms * 1e3, 1)
    print(json.dumps(res), flush=True)
