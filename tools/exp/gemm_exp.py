import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import gemm
from gemm_bench import timeit  # noqa
os.environ["ODW_GEMM_VARIANT"] = "big"
for name, M, N, K in [("fc6_fwd", 4000, 4096, 25088), ("fc6_dgrad", 4000, 25088, 4096), ("sq8192", 8192, 8192, 8192)]:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda")
    res = {"shape": name}
    for x in sys.argv[1].split(","):
        os.environ["ODW_GEMM_EXP"] = x
        out.zero_()
        ms = timeit(lambda: gemm.gemm_nt(a, b, M, N, K, out), iters=20)
        res["x" + x] = round(2.0 * M * N * K / ms / 1e9, 1)
        if x == "0": ref = out.clone()
        elif x in ("1", "6", "7", "8"): res["err" + x] = (out - ref).abs().max().item()
    print(json.dumps(res), flush=True)
