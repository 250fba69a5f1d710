"""Per-kernel timings of libodwscl.so on one MI355X (HIP events on the launch stream).

python tools/microbench.py [--out gpurun_out/microbench.json]
Reports ms per launch and the fraction of the HBM roofline on ALGORITHMIC bytes
(DESIGN.md gives the byte counts)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _C, synthetic  # noqa: E402
from od_wscl_amd.utils import rng  # noqa: E402

HBM_PEAK = 8.0e12


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/microbench.json")
    args = ap.parse_args()
    res = []

    def rec(name, ms, nbytes=None, flops=None, **kw):
        r = dict(kernel=name, ms=round(ms, 5))
        if nbytes:
            r["alg_GBps"] = round(nbytes / ms / 1e6, 1)
            r["hbm_frac"] = round(nbytes / (ms * 1e-3) / HBM_PEAK, 4)
        if flops:
            r["TFLOPs"] = round(flops / ms / 1e9, 2)
        r.update(kw)
        res.append(r)
        print(json.dumps(r), flush=True)

    for (P, S) in ((2000, 608), (4000, 800)):
        H = W = S // 8
        feat = torch.randn(1, 512, H, W, device="cuda")
        bx = synthetic.make_proposals(1234, 0, P, S - 8, S - 8)
        rois = torch.from_numpy(np.concatenate([np.zeros((P, 1), np.float32), bx], 1)).cuda()
        out_b = P * 512 * 49 * 4
        ms = timeit(lambda: _C.roi_pool_forward(feat, rois, 0.125, 7, 7))
        rec("roi_pool_fwd", ms, 2 * out_b + feat.numel() * 4, P=P, HW=H)
        out, arg = _C.roi_pool_forward(feat, rois, 0.125, 7, 7)
        g = torch.randn_like(out)
        ms = timeit(lambda: _C.roi_pool_backward(g, None, rois, arg, 0.125, 7, 7, 1, 512, H, W))
        rec("roi_pool_bwd", ms, 2 * out_b + feat.numel() * 4, P=P, HW=H)
        ms = timeit(lambda: _C.roi_align_forward(feat, rois, 0.125, 7, 7, 0), iters=5)
        rec("roi_align_fwd_sr0", ms, out_b + feat.numel() * 4, P=P, HW=H)
        ms = timeit(lambda: _C.roi_align_forward(feat, rois, 0.125, 7, 7, 2), iters=5)
        rec("roi_align_fwd_sr2", ms, out_b + feat.numel() * 4, P=P, HW=H)
        ms = timeit(lambda: _C.roi_align_backward(g, rois, 0.125, 7, 7, 1, 512, H, W, 2), iters=5)
        rec("roi_align_bwd_sr2", ms, out_b + feat.numel() * 4, P=P, HW=H)
        ms = timeit(lambda: _C.roi_align_backward(g, rois, 0.125, 7, 7, 1, 512, H, W, 0), iters=5)
        rec("roi_align_bwd_sr0", ms, out_b + feat.numel() * 4, P=P, HW=H)
        sc = torch.rand(P, device="cuda")
        bxd = rois[:, 1:].contiguous()
        ms = timeit(lambda: _C.nms_torchvision(bxd, sc, 0.1), iters=5)
        rec("nms_tv_0.1", ms, P=P)
        ms = timeit(lambda: _C.nms_torchvision(bxd, sc, 0.7), iters=5)
        rec("nms_tv_0.7", ms, P=P)
        E = torch.nn.functional.normalize(torch.randn(P, 128, device="cuda"), dim=1)
        ms = timeit(lambda: _C.pairwise_sim(E))
        rec("pairwise_sim", ms, P * 128 * 4 + P * P * 4, flops=2.0 * P * P * 128, P=P)
        ms = timeit(lambda: torch.mm(E, E.T))
        rec("torch_mm_EEt(rocBLAS, comparison)", ms, P * 128 * 4 + P * P * 4, flops=2.0 * P * P * 128, P=P)
        ms = timeit(lambda: _C.box_iou(bxd, bxd[:1]))
        rec("box_iou_Px1", ms, P=P)
    for N, ncls in ((185, 2), (1000, 20), (4000, 80)):
        F_ = torch.nn.functional.normalize(torch.randn(N, 128, device="cuda"), dim=1)
        y = torch.randint(0, ncls, (N,), device="cuda", dtype=torch.int32)
        w = torch.rand(N, device="cuda")
        ms = timeit(lambda: _C.supcon_v2(F_, y, w, 0.2, need_grad=False))
        rec("supcon_fwd", ms, flops=2.0 * N * N * 128, N=N)
        ms = timeit(lambda: _C.supcon_v2(F_, y, w, 0.2))
        rec("supcon_fwd_bwd", ms, flops=6.0 * N * N * 128, N=N)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
