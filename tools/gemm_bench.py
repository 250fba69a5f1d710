"""GEMM shapes of the ROI head: hand-written kernel variants vs hipBLASLt (torch.matmul bf16)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import gemm

def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

if __name__ != "__main__":
    shapes = []
else:
  shapes = [("fc6_fwd  M=4000", 4000, 4096, 25088), ("fc6_fwd  M=2000", 2000, 4096, 25088),
          ("fc6_dgrad", 4000, 25088, 4096), ("fc6_wgrad", 4096, 25088, 4032),
          ("fc7_fwd", 4000, 4096, 4096), ("fc6_fwd M=400 (K rows)", 400, 4096, 25088), ("predictor", 2000, 357, 4096),
          ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    k64 = (K + 63) // 64 * 64
    a = (torch.randn(M, k64, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, k64, device="cuda") * 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    outf = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * K
    res = {"shape": name, "M": M, "N": N, "K": K}
    for var in ("ring", "glds", "reg"):
        os.environ["ODW_GEMM_VARIANT"] = var
        ms = timeit(lambda: gemm.gemm_nt(a, b, M, N, K, out))
        res[var + "_TF"] = round(fl / ms / 1e9, 1); res[var + "_ms"] = round(ms, 4)
    os.environ["ODW_GEMM_VARIANT"] = "ring"
    ms = timeit(lambda: gemm.gemm_nt(a, b, M, N, K, outf))
    res["ring_f32out_TF"] = round(fl / ms / 1e9, 1)
    ms = timeit(lambda: torch.matmul(a[:, :K], b[:, :K].T))
    res["hipblaslt_TF"] = round(fl / ms / 1e9, 1); res["hipblaslt_ms"] = round(ms, 4)
    print(json.dumps(res), flush=True)
