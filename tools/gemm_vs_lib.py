"""The step's three fc6 products on the hand-written kernel and on torch.matmul (hipBLASLt), bf16 in / bf16 out."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import gemm
def t(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, (M, N, K) in (("fc6 fwd stacked", (4000, 4096, 25088)), ("fc6 dgrad", (2000, 25088, 4096)), ("fc6 wgrad", (4096, 25088, 2752)),
                        ("fc7 fwd", (4000, 4096, 4096)), ("8192^3", (8192, 8192, 8192))):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: gemm.gemm_nt(a, b, M, N, K, out))
    bt = b.t()
    ul = t(lambda: torch.matmul(a, bt, out=out))
    fl = 2.0 * M * N * K
    print("%-16s %5dx%5dx%5d  ours %7.1f us = %6.1f TF   hipBLASLt %7.1f us = %6.1f TF" % (name, M, N, K, us, fl / us / 1e6, ul, fl / ul / 1e6), flush=True)
