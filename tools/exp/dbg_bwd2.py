import os, sys
os.environ["ODW_BWD2"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from od_wscl_amd import engine, gemm
from od_wscl_amd.utils.device_rand import DeviceRand
dev = torch.device("cuda", 0)
orig = gemm.WgradBatch.flush
def flush(self, weight, tag=None):
    if self.dzt is not None:
        k = self.kpad
        print("FLUSH", tag, tuple(weight.shape), "rows", self.rows, "done", self.done, "kpad", k, "cap", self.dzt.shape[1], "split", self.split,
              "nan dzt", bool(torch.isnan(self.dzt[:, :k].float()).any()), "nan xt", bool(torch.isnan(self.xt[:, :k].float()).any()),
              "inf dzt", bool(torch.isinf(self.dzt[:, :k].float()).any()), "inf xt", bool(torch.isinf(self.xt[:, :k].float()).any()))
        off = 0
        for r, d in zip(self.rows, self.done):
            print("   block", off, r, d, "nan", bool(torch.isnan(self.dzt[:, off:off + r].float()).any()), bool(torch.isnan(self.xt[:, off:off + r].float()).any()))
            off += r
    return orig(self, weight, tag)
gemm.WgradBatch.flush = flush
orig_bs = gemm._backward_split
def bs(x32, y, weight, bias, cfg, dy, need_dx):
    print("BSPLIT", cfg[4], tuple(x32.shape), x32.dtype, x32.stride(), "nan x", bool(torch.isnan(x32).any()), "inf x", bool(torch.isinf(x32).any()),
          "nan dy", bool(torch.isnan(dy).any()), "y", None if y is None else (tuple(y.shape), bool(torch.isnan(y).any())), "grad_rows", cfg[5], "slot", cfg[6])
    return orig_bs(x32, y, weight, bias, cfg, dy, need_dx)
gemm._backward_split = bs
orig_sp = gemm._backward_single_plane
def sp(xb, y, weight, bias, cfg, dy, need_dx):
    print("BSINGLE", cfg[4], tuple(xb.shape), xb.dtype, "bwd2", cfg[0].bwd2, "slot", cfg[6], "grad_rows", cfg[5])
    return orig_sp(xb, y, weight, bias, cfg, dy, need_dx)
gemm._backward_single_plane = sp
cfg = bench.build_cfg(21)
step, info = engine.build_training_step(cfg, dev, dtype="bf16x2f", world=1, seed=cfg.SEED)
images, targets, rois = bench.synthetic_batch(1234, 2, 600, 2000, 21, dev)
for it in range(1):
    losses, _ = step(images, targets, rois, DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev))
    torch.cuda.synchronize()
    opt = step.optimizer
    for n, (o, k) in opt.slices.items():
        g = opt.flat_g[o:o + k]
        if torch.isnan(g).any():
            print("NaN grad:", n, int(torch.isnan(g).sum()), "of", k)
