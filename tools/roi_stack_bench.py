"""Fused ROIPool->stacked-operand kernels on the bench's shapes (P=2000, 76x76x512): fwd / bwd time, experiment modes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _lib as L, synthetic
lib = L.lib()
P, C, H, W = 2000, 512, 76, 76
feat = torch.randn(1, C, H, W, device="cuda").bfloat16().float()
bx = torch.from_numpy(synthetic.make_proposals(1234, 0, P, 600, 600)).cuda()
rois = torch.cat([torch.zeros(P, 1, device="cuda"), bx], dim=1).contiguous()
keep = (torch.rand(P, 49, device="cuda") > 0.1).float()
ks = keep.sum()
x = torch.empty(2 * P, C * 49, dtype=torch.bfloat16, device="cuda")
am = torch.empty(P, C * 49, dtype=torch.int16, device="cuda")
wsb = lib.odw_roi_pool_workspace(P, 7, 7); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
dx = torch.randn(2 * P, C * 49, device="cuda").bfloat16()
dfeat = torch.empty_like(feat)
def fwd():
    L.check(lib.odw_roi_pool_stack_forward(L.ptr(feat), L.ptr(rois), 0.125, 1, C, H, W, P, 7, 7, L.ptr(keep), L.ptr(ks), L.ptr(x),
                                           x.stride(0), L.ptr(am), L.ptr(ws), wsb, L.stream()), "fwd")
def bwd():
    L.check(lib.odw_roi_pool_stack_backward(L.ptr(dx), 0, dx.stride(0), L.ptr(am), L.ptr(rois), L.ptr(keep), L.ptr(ks), None, None, 0, 0,
                                            1, C, H, W, P, 7, 7, L.ptr(dfeat), L.stream()), "bwd")
def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
area = ((bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1]) / 64).mean().item()
print("mean ROI area %.0f cells" % area)
for mode in sys.argv[1].split(","):
    os.environ["ODW_RPS_MODE"] = mode
    print("mode", mode, "fwd %.1f us" % timeit(fwd), flush=True)
os.environ["ODW_RPS_MODE"] = "0"
fwd()
print("bwd %.1f us" % timeit(bwd))
def bwd_skip():
    L.check(lib.odw_roi_pool_stack_backward(L.ptr(dx), 0, dx.stride(0), L.ptr(am), L.ptr(rois), L.ptr(keep), L.ptr(ks), None, None, 0, 1,
                                            1, C, H, W, P, 7, 7, L.ptr(dfeat), L.stream()), "bwd")
print("bwd skip_clean %.1f us" % timeit(bwd_skip))
nhwc = feat.permute(0, 2, 3, 1).contiguous().bfloat16()
wsn = lib.odw_roi_pool_stack_nhwc_workspace(P, 1, C, H, W); ws2 = torch.empty(wsn, dtype=torch.uint8, device="cuda")
def fwd_nhwc():
    L.check(lib.odw_roi_pool_stack_forward_nhwc(L.ptr(nhwc), L.ptr(rois), 0.125, 1, C, H, W, P, L.ptr(keep), L.ptr(ks), L.ptr(x),
                                                x.stride(0), L.ptr(am), L.ptr(ws2), wsn, L.stream()), "fwd")
print("fwd nhwc %.1f us" % timeit(fwd_nhwc))
