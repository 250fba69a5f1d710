"""torch.profiler view of the host side of one training step (CPU self/total time per op, incl. autograd nodes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ODW_NO_TIMER"] = os.environ.get("ODW_NO_TIMER", "0")
import bench
from od_wscl_amd import engine
from od_wscl_amd.utils.device_rand import DeviceRand
from torch.profiler import profile, ProfilerActivity
cfg = bench.build_cfg(21); dev = torch.device("cuda", 0)
step, info = engine.build_training_step(cfg, dev, dtype="bf16", world=1, backend="hip")
images, targets, rois = bench.synthetic_batch(1234, 0, 600, 2000, 21, dev)
for it in range(5):
    step(images, targets, rois, DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev))
torch.cuda.synchronize()
N = 5
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for it in range(5, 5 + N):
        step(images, targets, rois, DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev))
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.cpu_time_total)[:45]
print("%-70s %8s %10s %10s" % ("op", "calls/st", "total_us/st", "self_us/st"))
for e in rows:
    print("%-70s %8.1f %10.1f %10.1f" % (e.key[:70], e.count / N, e.cpu_time_total / N, e.self_cpu_time_total / N))
print("--- by self time")
for e in sorted(ev, key=lambda e: -e.self_cpu_time_total)[:30]:
    print("%-70s %8.1f %10.1f %10.1f" % (e.key[:70], e.count / N, e.cpu_time_total / N, e.self_cpu_time_total / N))
