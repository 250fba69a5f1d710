"""Host-side cost of the launch wrappers (the loss zone of the step is host-bound): calls per second of gemm.gemm_nt and of
its parts on shapes whose GPU time is far below the host time."""
import os, sys, time, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _lib as L, gemm
lib = L.lib()
M, N, K = 64, 128, 128
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def rate(fn, n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    h = time.perf_counter() - t
    torch.cuda.synchronize()
    return h / n * 1e6
print("gemm_nt (python wrapper)        %.1f us/call" % rate(lambda: gemm.gemm_nt(a, b, M, N, K, out)))
var = ctypes.c_int(0)
print("  workspace query               %.1f us" % rate(lambda: lib.odw_gemm_nt_bf16_workspace(M, N, K, K, K, L.ptr(out), N, 1, ctypes.byref(var))))
print("  torch.empty(1 KB)             %.1f us" % rate(lambda: torch.empty(1024, dtype=torch.uint8, device="cuda")))
print("  L.ptr x6 + L.stream           %.1f us" % rate(lambda: (L.ptr(a), L.ptr(b), L.ptr(out), L.ptr(None), L.ptr(None), L.ptr(None), L.stream())))
st = L.stream()
print("  launch call alone             %.1f us" % rate(lambda: lib.odw_gemm_nt_bf16_ws(L.ptr(a), K, L.ptr(b), K, M, N, K, L.ptr(out), N, 1, None, 0, 1.0, 0.0, 0, None, None, None, 0, None, 0, st)))
x = torch.randn(256, 4096, device="cuda").bfloat16().requires_grad_(True)
from od_wscl_amd.layers.linear import Linear
lin = Linear(4096, 4096).cuda()
def fwd_bwd():
    y = lin.fused(x, relu=True)
    y.backward(y)
print("Linear.fused fwd+bwd (256 rows) %.1f us/call (host; GPU ~ 60 us)" % rate(fwd_bwd, 300))
