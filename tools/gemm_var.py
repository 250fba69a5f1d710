"""A/B of GEMM kernel variants (ODW_GEMM_VARIANT) on the ROI-head shapes, same box, fp32 or bf16 output."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import gemm
from gemm_bench import timeit  # noqa

variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["ring", "glds", "big"]
shapes = [("fc6_fwd", 4000, 4096, 25088), ("fc6_dgrad", 4000, 25088, 4096), ("fc6_wgrad", 4096, 25088, 4032),
          ("fc7_fwd", 4000, 4096, 4096), ("sq8192", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    k64 = (K + 63) // 64 * 64
    a = (torch.randn(M, k64, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, k64, device="cuda") * 0.5).bfloat16()
    fl = 2.0 * M * N * K
    ref = torch.matmul(a[:, :K].float()[:256], b[:, :K].float().T)
    for dt in (torch.float32, torch.bfloat16):
        out = torch.empty(M, N, device="cuda", dtype=dt)
        res = {"shape": name, "out": str(dt)[6:]}
        for var in variants:
            os.environ["ODW_GEMM_VARIANT"] = var
            out.zero_()
            ms = timeit(lambda: gemm.gemm_nt(a, b, M, N, K, out), iters=20)
            err = (out[:256].float() - ref).abs().max().item() / ref.abs().max().item()
            res[var] = round(fl / ms / 1e9, 1)
            if err > 1e-2: res[var + "_ERR"] = err
        ms = timeit(lambda: torch.matmul(a[:, :K], b[:, :K].T), iters=20)
        res["hipblaslt"] = round(fl / ms / 1e9, 1)
        print(json.dumps(res), flush=True)
