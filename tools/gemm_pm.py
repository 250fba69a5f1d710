"""A/B of the tile-order group height (ODW_GEMM_PM) on the ROI-head GEMM shapes, same box, interleaved."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import gemm
from gemm_bench import timeit  # noqa

shapes = [("fc6_fwd", 4000, 4096, 25088), ("fc6_dgrad", 4000, 25088, 4096), ("fc6_wgrad", 4096, 25088, 4032),
          ("fc7_fwd", 4000, 4096, 4096)]
for name, M, N, K in shapes:
    k64 = (K + 63) // 64 * 64
    a = (torch.randn(M, k64, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, k64, device="cuda") * 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for var in ("ring", "glds"):
        os.environ["ODW_GEMM_VARIANT"] = var
        res = {"shape": name, "var": var}
        for pm in (1000, 2, 4, 8, 16):
            os.environ["ODW_GEMM_PM"] = str(pm)
            ms = timeit(lambda: gemm.gemm_nt(a, b, M, N, K, out), iters=20)
            res["pm%d" % pm] = round(fl / ms / 1e9, 1)
        print(json.dumps(res), flush=True)
