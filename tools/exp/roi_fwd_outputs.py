"""The fused ROIPool forward of the bf16x2f step (roi_pool_stack_fwd_nhwc_f32, pair layout): what each output costs."""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from od_wscl_amd import _lib as L, synthetic
lib = L.lib()
P, C, H, W = 2000, 512, 76, 76
K = C * 49
nhwc32 = torch.randn(1, H, W, C, device="cuda").relu().contiguous()
bx = torch.from_numpy(synthetic.make_proposals(1234, 0, P, 600, 600)).cuda()
rois = torch.cat([torch.zeros(P, 1, device="cuda"), bx], dim=1).contiguous().float()
keep = (torch.rand(P, 49, device="cuda") > 0.1).float(); ks = keep.sum()
planes = torch.empty(2 * P, K, dtype=torch.bfloat16, device="cuda")
cm = torch.empty(P, 2 * K, dtype=torch.bfloat16, device="cuda")
pooled = torch.empty(P, K, dtype=torch.float32, device="cuda")
am = torch.empty(P, K, dtype=torch.int16, device="cuda")
wsb = lib.odw_roi_pool_stack_nhwc_f32_workspace(P, 1, C, H, W); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
pat = (ctypes.c_int * 1)(0)
def run(with_pooled, with_cm):
    L.check(lib.odw_roi_pool_stack_forward_nhwc_f32_cm(L.ptr(nhwc32), L.ptr(rois), 0.125, 1, C, H, W, P, L.ptr(keep), L.ptr(ks),
                                                       ctypes.cast(pat, ctypes.c_void_p), 1, L.ptr(planes), planes.stride(0), K,
                                                       L.ptr(pooled) if with_pooled else None, L.ptr(am),
                                                       L.ptr(cm) if with_cm else None, 2 * K if with_cm else 0, K if with_cm else 0,
                                                       L.ptr(ws), wsb, L.stream()), "fwd")
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for wp, wc in ((True, True), (False, True), (True, False), (False, False)):
    print("pooled32 %d  cell-major planes %d : %.1f us (incl. the ordinal pre-pass and the bin table)" % (wp, wc, timeit(lambda: run(wp, wc))))
