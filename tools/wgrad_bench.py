"""Convolution weight gradient at the bench's layer shapes: the NT chain (dZ^T from linear_bwd_prep + transposed im2col +
split-K GEMM + reduce-and-unpack) against the TN form (odw_conv_wgrad_tn: K-major operands read with ds_read_b64_tr_b16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _lib as L
lib = L.lib()
st = L.stream()
zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
r64 = lambda n: (n + 63) // 64 * 64

def timeit(fn, iters=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / iters

for cin, cout, dil, h in [(128, 256, 1, 152), (256, 256, 1, 152), (256, 512, 1, 76), (512, 512, 1, 76), (512, 512, 2, 76)]:
    m = h * h; m64 = r64(m)
    x = torch.randn(m, cin, device="cuda").bfloat16()
    dz = torch.randn(m, cout, device="cuda").bfloat16()
    dzc = torch.empty(m, r64(cout), dtype=torch.bfloat16, device="cuda")
    dzt = torch.empty(cout, m64, dtype=torch.bfloat16, device="cuda")
    db = torch.zeros(cout, device="cuda")
    colt = torch.empty(9 * cin, m64, dtype=torch.bfloat16, device="cuda")
    dw = torch.empty(cout, cin, 3, 3, device="cuda")
    wsb = lib.odw_conv_wgrad_workspace(cout, cin, m, m64, m64); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    wsb2 = lib.odw_conv_wgrad_tn_workspace(cout, cin, m); ws2 = torch.empty(wsb2, dtype=torch.uint8, device="cuda")
    def prep():
        L.check(lib.odw_linear_bwd_prep(L.ptr(dz), 0, cout, None, 0, m, cout, 1.0, L.ptr(dzc), dzc.stride(0), L.ptr(dzt), m64, L.ptr(db), st), "prep")
    def nt():
        prep()
        L.check(lib.odw_im2col_t_bf16(L.ptr(x), m, h, h, cin, dil, L.ptr(colt), m64, st), "im2col")
        L.check(lib.odw_conv_wgrad_nt(L.ptr(dzt), m64, L.ptr(colt), m64, cout, cin, cin, m, L.ptr(dw), 0, L.ptr(ws), wsb, st), "nt")
    def tn():
        L.check(lib.odw_conv_wgrad_tn(L.ptr(dz), cout, L.ptr(x), m, h, h, cin, dil, cout, cin, L.ptr(dw), 0, L.ptr(zero), L.ptr(ws2), wsb2, st), "tn")
    t_prep, t_nt, t_tn = timeit(prep), timeit(nt), timeit(tn)
    fl = 2.0 * m * cout * 9 * cin
    print("%4d->%4d dil %d %3dx%-3d  prep %5.1f us | NT chain (prep + im2col_t + GEMM + reduce) %6.1f us | TN %6.1f us = %5.0f TF  (splits %d)" %
          (cin, cout, dil, h, h, t_prep, t_nt, t_tn, fl / t_tn / 1e6, wsb2 // (cout * 9 * cin * 4)))
