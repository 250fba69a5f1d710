"""Per-layer time of the implicit-GEMM convolutions (forward and input-gradient) at the bench's 608x608 input."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _lib as L
from od_wscl_amd.modeling.backbone.vgg16_hip import _r64
lib = L.lib()
zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
H = 608
layers = [(3, 64, 1, H), (64, 64, 1, H), (64, 128, 1, H // 2), (128, 128, 1, H // 2), (128, 256, 1, H // 4), (256, 256, 1, H // 4),
          (256, 256, 1, H // 4), (256, 512, 1, H // 8), (512, 512, 1, H // 8), (512, 512, 1, H // 8), (512, 512, 2, H // 8),
          (512, 512, 2, H // 8), (512, 512, 2, H // 8)]
tot = 0
for li, (cin, cout, dil, h) in enumerate(layers):
    cp = max(8, 1 << (cin - 1).bit_length())
    m = h * h
    x = torch.randn(m, cp, device="cuda").bfloat16()
    wk = torch.randn(cout, _r64(9 * cp), device="cuda").bfloat16()
    y = torch.empty(m, cout, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(cout, device="cuda")
    wsb = lib.odw_conv3x3_workspace_hw(m, h, h, cp, cout, dil)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    def run():
        L.check(lib.odw_conv3x3_nhwc_bf16_ws(L.ptr(x), m, h, h, cp, dil, 0, L.ptr(wk), wk.stride(0), cout, L.ptr(y), cout, 1,
                                             L.ptr(bias), 1, None, 0, L.ptr(zero), L.ptr(ws) if wsb else None, wsb, L.stream()), "conv")
    for _ in range(3): run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    fl = 2.0 * m * cout * 9 * cin
    tot += us
    print("conv%-2d  %4d->%4d dil %d  %3dx%-3d  M=%6d  %7.1f us  %6.1f TF (real FLOPs)  tiles %d splits %d" %
          (li, cin, cout, dil, h, h, m, us, fl / us / 1e6, ((m + 127) // 128) * ((cout + 127) // 128), wsb // (m * cout * 4)))
print("forward total %.1f us" % tot)
