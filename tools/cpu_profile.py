"""cProfile of the host side of the training step (the step is host-bound: find the Python hot spots)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from od_wscl_amd import engine
from od_wscl_amd.utils.device_rand import DeviceRand
cfg = bench.build_cfg(21); dev = torch.device("cuda", 0)
step, info = engine.build_training_step(cfg, dev, dtype=(sys.argv[1] if len(sys.argv) > 1 else "bf16x2f"), world=1, backend="hip")
images, targets, rois = bench.synthetic_batch(1234, 0, 600, 2000, 21, dev)
for it in range(5):
    step(images, targets, rois, DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev))
torch.cuda.synchronize()
if os.environ.get("ODW_PROFILE_BACKWARD") == "1":      # run autograd on THIS thread, so that cProfile sees the backward functions
    torch.autograd.set_multithreading_enabled(False)
    for it in range(3):
        step(images, targets, rois, DeviceRand(1234, first_stream=(1 << 20) + ((20 + it) << 12), device=dev))
    torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for it in range(5, 15):
    step(images, targets, rois, DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
