"""One backbone convolution in isolation (for tools/pmc_conv.sh): ODW_CONV_LAYER = conv3 | conv4 | conv5 | conv4_dgrad |
conv2 picks the VGG16-OICR layer at the bench's 608x608 input; 8 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _lib as L
from od_wscl_amd.modeling.backbone.vgg16_hip import _r64
lib = L.lib()
LAYERS = {"conv2": (128, 128, 1, 304, 0), "conv3": (256, 256, 1, 152, 0), "conv4": (512, 512, 1, 76, 0),
          "conv5": (512, 512, 2, 76, 0), "conv4_dgrad": (512, 512, 1, 76, 1), "conv3_dgrad": (256, 256, 1, 152, 1)}
cin, cout, dil, h, mirror = LAYERS[os.environ.get("ODW_CONV_LAYER", "conv4")]
cin = int(os.environ.get("ODW_CONV_CIN", cin))          # (fixed-overhead fits: time against the K of one layer shape)
T = int(os.environ.get("ODW_CONV_PLANES", "1"))         # 3 = the forward of the "bf16x2f" mode: three plane blocks, fp32 out
cin_real = cin
cin *= T
m = h * h
zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
if os.environ.get("ODW_CONV_P2") == "1":
    # round 5: the two-plane forward (conv3x3_halo2_kernel): operand [hi C | mid C] per pixel, weights [hi 32 | mid 32] per
    # tap and 32-channel block; ODW_CONV_P2_OUT=planes: the epilogue writes the next layer's planes instead of fp32
    c = cin_real
    planes = os.environ.get("ODW_CONV_P2_OUT") == "planes"
    x = torch.randn(m, 2 * c, device="cuda").bfloat16()
    wk = (torch.randn(cout, _r64(18 * c), device="cuda") * 0.05).bfloat16()
    y = torch.empty((m, 2 * cout), device="cuda", dtype=torch.bfloat16) if planes else torch.empty((m, cout), device="cuda")
    bias = torch.zeros(cout, device="cuda")
    wsb = lib.odw_conv3x3_planes2_workspace(m, h, h, c, cout)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    for _ in range(8):
        L.check(lib.odw_conv3x3_planes2_ws(L.ptr(x), x.stride(0), m, h, h, c, dil, L.ptr(wk), wk.stride(0), cout, L.ptr(y), y.stride(0),
                                           1 if planes else 0, L.ptr(bias), 1, L.ptr(zero), L.ptr(ws) if wsb else None, wsb, L.stream()),
                "conv planes2")
    torch.cuda.synchronize()
    print("layer", os.environ.get("ODW_CONV_LAYER", "conv4"), "two-plane forward: M", m, "N", cout, "C", c, "GFLOP issued", 2e-9 * m * cout * 9 * c * 3,
          "algorithmic", 2e-9 * m * cout * 9 * c, "split bytes", wsb, "out", "planes" if planes else "fp32")
    sys.exit(0)
x = torch.randn(m, cin, device="cuda").bfloat16()
wk = (torch.randn(cout, _r64(9 * cin), device="cuda") * 0.05).bfloat16()
y = torch.empty(m, cout, device="cuda", dtype=torch.bfloat16 if T == 1 else torch.float32)
bias = torch.zeros(cout, device="cuda")
wsb = lib.odw_conv3x3_workspace_hw(m, h, h, cin, cout, dil)
ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
for _ in range(8):
    L.check(lib.odw_conv3x3_nhwc_bf16_ws(L.ptr(x), m, h, h, cin, dil, mirror, L.ptr(wk), wk.stride(0), cout, L.ptr(y), cout, 1 if T == 1 else 0,
                                         L.ptr(bias), 0 if mirror else 1, None, 0, L.ptr(zero), L.ptr(ws) if wsb else None, wsb,
                                         L.stream()), "conv")
torch.cuda.synchronize()
print("layer", os.environ.get("ODW_CONV_LAYER", "conv4"), "M", m, "N", cout, "K", 9 * cin, "GFLOP issued", 2e-9 * m * cout * 9 * cin, "plane blocks", T, "split bytes", wsb)
