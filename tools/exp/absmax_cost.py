"""fc6's input-gradient product (M = 2000, N = 25088, K = 4096, fp32 out) with and without the absmax epilogue, and the
stand-alone pre-pass it replaces.  python tools/exp/absmax_cost.py"""
import torch
from od_wscl_amd import _lib as L, gemm, precision

precision.set_precision("bf16")
M, N, K = 2000, 25088, 4096
a = (torch.randn(M, K, device="cuda") * 0.1).bfloat16()
b = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
out = torch.empty(M, N, device="cuda")
word = torch.zeros(16, dtype=torch.int32, device="cuda")


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rnd in range(3):
    t0 = timed(lambda: gemm.gemm_nt(a, b, M, N, K, out))
    t1 = timed(lambda: (word.zero_(), gemm.gemm_nt(a, b, M, N, K, out, absmax=word)))
    t2 = timed(lambda: word.zero_())
    print("plain %.1f us   with absmax (+ zeroing the word) %.1f us   zeroing alone %.1f us" % (t0, t1, t2))
