"""A plain Linear of the bf16x2f forward (fc7, Sim_Net's first layer: K = 4096) on the two-plane kernel of fc6 (gemm_nt_cm_kernel
with ONE cell: operands [hi | mid], three plane products per K step off one load of each plane) against the K-concatenated form
the step uses ([hi hi mid] x [hi mid hi], K' = 3K on the 256 x 256 kernel).  python tools/exp/cm_plain_linear.py"""
import torch
from od_wscl_amd import gemm, precision

precision.set_precision("bf16x2f")
dev = torch.device("cuda:0")


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def planes(x):
    hi = x.bfloat16()
    mid = (x - hi.float()).bfloat16()
    return hi, mid


for (M, N, K, what) in ((4000, 4096, 4096, "fc7 over the stacked pass"), (2000, 4096, 4096, "Sim_Net layer 0 over the clean rows"),
                        (896, 4096, 4096, "views, three labels"), (320, 4096, 4096, "views, one label")):
    x = torch.randn(M, K, device=dev) * 0.5
    w = torch.randn(N, K, device=dev) * 0.02
    xh, xm = planes(x)
    wh, wm = planes(w)
    a3 = torch.cat([xh, xh, xm], 1).contiguous()
    b3 = torch.cat([wh, wm, wh], 1).contiguous()
    a2 = torch.cat([xh, xm], 1).contiguous()
    b2 = torch.cat([wh, wm], 1).contiguous()
    bias = torch.randn(N, device=dev)
    o3 = torch.empty(M, N, device=dev)
    o2 = torch.empty(M, N, device=dev)
    t3 = timed(lambda: gemm.gemm_nt(a3, b3, M, N, 3 * K, o3, bias=bias, relu=True, planes=3))
    t2 = timed(lambda: gemm.gemm_nt_cm(a2, b2, M, N, K, 1, o2, bias=bias, relu=True))
    ref = torch.relu(x.double() @ w.double().t() + bias.double())
    e3 = ((o3.double() - ref).abs().max() / ref.abs().max()).item()
    e2 = ((o2.double() - ref).abs().max() / ref.abs().max()).item()
    print("%-38s M=%d: K-concatenated %.1f us (err %.1e)   two-plane kernel %.1f us (err %.1e)   max |diff| %.2e"
          % (what, M, t3, e3, t2, e2, (o3 - o2).abs().max().item()))
