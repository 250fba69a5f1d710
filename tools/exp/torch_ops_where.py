"""Which lines of the package issue the torch kernels of a step (copies, fills, gathers, cats): aten op x first package frame.
(TorchDispatchMode + traceback: the profiler's with_stack gives no Python frames in this build.)"""
import os, sys, collections, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from od_wscl_amd import engine
from od_wscl_amd.utils.device_rand import DeviceRand
from torch.utils._python_dispatch import TorchDispatchMode
cfg = bench.build_cfg(21); dev = torch.device("cuda", 0)
step, info = engine.build_training_step(cfg, dev, dtype=os.environ.get("ODW_DTYPE", "bf16x2f"), world=1, backend="hip")
img = int(os.environ.get("ODW_IMG", "1"))
images, targets, rois = bench.synthetic_batch(1234, img, 600, 2000, 21, dev)
for it in range(6):
    step(images, targets, rois, DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev))
torch.cuda.synchronize()
SKIP = ("view", "reshape", "as_strided", "slice", "select", "detach", "alias", "t.default", "transpose", "permute", "unsqueeze",
        "squeeze", "expand", "narrow", "_unsafe_view", "empty", "lift_fresh", "is_", "size", "stride", "_local_scalar", "item",
        "unbind", "split", "chunk", "record_stream", "resize_", "set_")
cnt = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            on_gpu = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
            if on_gpu or "zeros" in name or "full" in name:
                where = "?"
                for fr in reversed(traceback.extract_stack()):
                    if "od_wscl_amd" in fr.filename and "tools/exp" not in fr.filename:
                        where = "%s:%d %s" % (fr.filename.replace(ROOT + "/", ""), fr.lineno, (fr.line or "")[:90])
                        break
                cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))
N = 2
with Mode():
    for it in range(6, 6 + N):
        step(images, targets, rois, DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev))
torch.cuda.synchronize()
for (name, where), c in sorted(cnt.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%5.1f  %-28s %s" % (c / N, name.replace("aten.", ""), where[:170]))
