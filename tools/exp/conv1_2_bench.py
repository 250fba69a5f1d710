import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from od_wscl_amd import _lib as L
from od_wscl_amd.modeling.backbone.vgg16_hip import _r64
lib = L.lib()
cin, cout, dil, h = 64 * 3, 64, 1, 608
m = h * h
zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
x = torch.randn(m, cin, device="cuda").bfloat16()
wk = (torch.randn(cout, _r64(9 * cin), device="cuda") * 0.05).bfloat16()
y = torch.empty(m, cout, device="cuda", dtype=torch.float32)
bias = torch.zeros(cout, device="cuda")
wsb = lib.odw_conv3x3_workspace_hw(m, h, h, cin, cout, dil)
ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
f = lambda: L.check(lib.odw_conv3x3_nhwc_bf16_ws(L.ptr(x), m, h, h, cin, dil, 0, L.ptr(wk), wk.stride(0), cout, L.ptr(y), cout, 0, L.ptr(bias), 1, None, 0, L.ptr(zero), L.ptr(ws) if wsb else None, wsb, L.stream()), "conv")
for _ in range(5): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): f()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
print("conv1_2 (3 plane blocks, fp32 out) %s: %.1f us = %.2f PF useful" % (os.environ.get("ODW_CONV_NO_N64", "N64"), us, 2.0 * m * 64 * 9 * cin / us / 1e9))
