"""Which allocations still make the caching allocator go to hipMalloc after the bench's warm-up (a ~1 ms bump in per_step_ms)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from od_wscl_amd import engine
from od_wscl_amd.utils.device_rand import DeviceRand
dev = torch.device("cuda", 0)
cfg = bench.build_cfg(21, "vgg16", "ROIPool", "fp32")
images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 600, 2000, 21, dev, n_images=1)
step, _ = engine.build_training_step(cfg, dev, dtype="bf16x2f", world=1, seed=cfg.SEED)
engine.kernel_timer.active = False
for it in range(5): step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20) + (it << 12), device=dev))
torch.cuda.synchronize()
torch.cuda.memory._record_memory_history(max_entries=200000, stacks="python")
n0 = torch.cuda.memory_stats(dev)["num_device_alloc"]
for it in range(5, 45):
    step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20) + (it << 12), device=dev))
    torch.cuda.synchronize()
    n1 = torch.cuda.memory_stats(dev)["num_device_alloc"]
    if n1 != n0: print("step", it, "device allocs", n1 - n0); n0 = n1
snap = torch.cuda.memory._snapshot()
for tr in snap["device_traces"]:
    for e in tr:
        if e["action"] == "segment_alloc":
            fr = [f for f in e.get("frames", []) if "od_wscl_amd" in f["filename"]][:3]
            print("segment_alloc %8.1f MB  <- %s" % (e["size"] / 2**20, " <- ".join("%s:%d" % (os.path.basename(f["filename"]), f["line"]) for f in fr)))
