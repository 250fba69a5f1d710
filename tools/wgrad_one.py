"""One convolution weight gradient (odw_conv_wgrad_tn) at a bench layer shape, a few launches: the target of tools/pmc_wgrad.sh."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _lib as L
lib = L.lib(); st = L.stream()
cin, cout, dil, h = {"conv3": (256, 256, 1, 152), "conv4": (512, 512, 1, 76), "conv5": (512, 512, 2, 76)}[os.environ.get("ODW_WGRAD_LAYER", "conv4")]
m = h * h
zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
x = torch.randn(m, cin, device="cuda").bfloat16(); dz = torch.randn(m, cout, device="cuda").bfloat16()
dw = torch.empty(cout, cin, 3, 3, device="cuda")
wsb = lib.odw_conv_wgrad_tn_workspace(cout, cin, m); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
for _ in range(6):
    L.check(lib.odw_conv_wgrad_tn(L.ptr(dz), cout, L.ptr(x), m, h, h, cin, dil, cout, cin, L.ptr(dw), 0, L.ptr(zero), L.ptr(ws), wsb, st), "tn")
torch.cuda.synchronize()
print("done", cin, cout, dil, h)
