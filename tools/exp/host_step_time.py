"""How long the HOST spends inside one call of the training step (it returns with GPU work still queued) against the
wall-clock per step: host time ~ wall time means the step is host-bound somewhere."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from od_wscl_amd import engine
from od_wscl_amd.utils.device_rand import DeviceRand
dev = torch.device("cuda", 0)
cfg = bench.build_cfg(21, "vgg16", "ROIPool", "fp32")
images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 600, 2000, 21, dev, n_images=1)
step, _ = engine.build_training_step(cfg, dev, dtype="bf16x2f", world=1, seed=cfg.SEED)
engine.kernel_timer.active = False
m = step.model
phases = {}
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); phases.setdefault(tag, []).append(time.perf_counter() - t); return r
    setattr(obj, name, g)
wrap(step.optimizer, "step", "opt.step")
wrap(step.optimizer, "begin_step", "begin_step")
wrap(m, "forward", "model.forward")
host, N = [], 30
for it in range(8): step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20) + (it << 12), device=dev))
torch.cuda.synchronize(); phases.clear()
t0 = time.perf_counter()
for it in range(N):
    t = time.perf_counter()
    step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20) + ((8 + it) << 12), device=dev))
    host.append(time.perf_counter() - t)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
print("wall %.3f ms/step   host inside step() %.3f ms (median %.3f)" % (wall * 1e3, np.mean(host) * 1e3, np.median(host) * 1e3))
for k, v in phases.items(): print("  %-14s %.3f ms" % (k, np.mean(v) * 1e3))
