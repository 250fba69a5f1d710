"""Every timed region of one step (GEMMs, convolutions, pooling) by (region, launch shape): launches per step and average time."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from od_wscl_amd import engine
from od_wscl_amd.utils.device_rand import DeviceRand
cfg = bench.build_cfg(21); dev = torch.device("cuda", 0)
step, info = engine.build_training_step(cfg, dev, dtype=os.environ.get("ODW_DTYPE", "bf16x2f"), world=1, backend="hip")
img = int(os.environ.get("ODW_IMG", "1"))
images, targets, rois = bench.synthetic_batch(1234, img, 600, 2000, 21, dev)
kt = engine.kernel_timer
kt.active = False
for it in range(8):
    step(images, targets, rois, DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev))
torch.cuda.synchronize()
N = 4
kt.reset(prealloc=2 * 200 * N); kt.active = True
for it in range(8, 8 + N):
    step(images, targets, rois, DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev))
torch.cuda.synchronize()
rows = collections.OrderedDict()
for name, recs in kt.records.items():
    for r in recs:
        k = (name, r[5])
        a = rows.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += r[0].elapsed_time(r[1]); a[2] += r[2]
tot = 0.0
for (name, shape), (c, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if not name.startswith("layer/"):
        tot += ms / N
    print("%5.2f x %8.1f us = %7.3f ms/step  %6.0f TF/s  %-52s %s" % (c / N, ms / c * 1e3, ms / N, fl / ms / 1e9 if ms else 0, name[:52], shape))
print("total (without layer/ views): %.3f ms/step" % tot)
