"""fc6-shaped GEMM timed with HIP events per launch: back-to-back (operands warm in the MALL/L2) vs after a
600 MB memset (cold, as inside the training step) vs with the fused ReLU+dropout epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import gemm
M, N, K = 4000, 4096, 25088
a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
junk = torch.empty(600 << 20, dtype=torch.uint8, device="cuda")
def run(flush, epi, n=12):
    ts = []
    for i in range(n):
        if flush: junk.fill_(i)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        if epi: gemm.gemm_nt(a, b, M, N, K, out, bias=bias, relu=True, drop_p=0.5, segs=[(0, 1, 2), (2000, 3, 4)])
        else: gemm.gemm_nt(a, b, M, N, K, out)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts = sorted(ts[2:]); return ts[len(ts) // 2]
for flush in (False, True):
    for epi in (False, True):
        ms = run(flush, epi)
        print("flush=%s epilogue=%s  %.4f ms  %.1f TF" % (flush, epi, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
