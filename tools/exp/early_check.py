"""early_backward on / off: gradients after one step and losses of the second step, each configuration twice (noise level)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from od_wscl_amd import engine
from od_wscl_amd.utils.device_rand import DeviceRand
dev = torch.device("cuda", 0)
os.environ["ODW_NO_TIMER"] = "1"
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
res = []
for early in (True, False, True, False):
    os.environ["ODW_NO_EARLY_BWD"] = "0" if early else "1"
    cfg = bench.build_cfg(21)
    step, _ = engine.build_training_step(cfg, dev, dtype=dtype, world=1, seed=cfg.SEED, backend="hip")
    images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 224, 150, 21, dev)
    opt = step.optimizer
    l1, _ = step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20), device=dev))
    torch.cuda.synchronize()
    g = {n: opt.flat_g[o:o + k].double().norm().item() for n, (o, k) in opt.slices.items()}
    gs = {n: opt.flat_g[o:o + k].double().sum().item() for n, (o, k) in opt.slices.items()}
    l2, _ = step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20) + (1 << 12), device=dev))
    torch.cuda.synchronize()
    res.append((early, {k: float(v) for k, v in l1.items()}, g, gs, {k: float(v) for k, v in l2.items()}))
    del step, opt
base = res[1]
for early, l1, g, gs, l2 in res:
    worst = max(((abs(g[n] - base[2][n]) / max(base[2][n], 1e-12)), n) for n in g)
    print("early" if early else "plain", "l1 %.6f" % sum(l1.values()), "l2", {k: round(v, 5) for k, v in l2.items()})
    print("    worst grad-norm deviation vs first plain run: %.3e at %s" % worst)
    bad = sorted(((abs(g[n] - base[2][n]) / max(base[2][n], 1e-12)), n) for n in g)[-6:]
    print("    ", [(n, "%.2e" % d) for d, n in bad])
