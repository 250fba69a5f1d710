"""What a shared clean + DropBlock fc6 forward could gain: the stacked pass (M = 2P) on the default kernel against
the SAME kernels over the P clean rows only (ring / big / planner's choice), K' = 3 x 25088, fp32 out."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from od_wscl_amd import gemm
P, N, K = 2000, 4096, 25088 * 3
a = (torch.randn(2 * P, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
bias = torch.randn(N, device="cuda")
def t(M, segs, iters=6):
    out = torch.empty(M, N, device="cuda")
    for _ in range(2): gemm.gemm_nt(a, b, M, N, K, out, bias=bias, relu=True, drop_p=0.5, segs=segs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): gemm.gemm_nt(a, b, M, N, K, out, bias=bias, relu=True, drop_p=0.5, segs=segs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12
which = os.environ.get("ODW_GEMM_VARIANT", "auto")
print("variant %-5s  M=4000: %.3f ms %.0f TF   M=2000: %.3f ms %.0f TF" % ((which,) + t(2 * P, [(0, 1, 2), (P, 3, 4)]) + t(P, [(0, 1, 2)])), flush=True)
