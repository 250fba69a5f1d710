"""plain backward with WgradBatch.reserve = 0 / 1024: does the leading dimension of the batch buffers change the numbers?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from od_wscl_amd import engine, gemm
from od_wscl_amd.utils.device_rand import DeviceRand
dev = torch.device("cuda", 0)
os.environ["ODW_NO_TIMER"] = "1"
os.environ["ODW_NO_EARLY_BWD"] = "1"
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
res = []
for reserve, fuse in ((0, "0"), (1024, "0"), (0, "1"), (1024, "1"), (0, "0")):
    os.environ["ODW_NO_PRED_FUSE"] = fuse
    cfg = bench.build_cfg(21)
    step, _ = engine.build_training_step(cfg, dev, dtype=dtype, world=1, seed=cfg.SEED, backend="hip")
    gemm.WgradBatch.reserve = reserve
    images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 224, 150, 21, dev)
    opt = step.optimizer
    l1, _ = step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20), device=dev))
    torch.cuda.synchronize()
    g = {n: opt.flat_g[o:o + k].double().clone() for n, (o, k) in opt.slices.items()}
    l2, _ = step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20) + (1 << 12), device=dev))
    torch.cuda.synchronize()
    res.append((reserve, fuse, g, {k: float(v) for k, v in l2.items()}))
    del step, opt
base = res[0][2]
for reserve, fuse, g, l2 in res:
    dev_ = sorted(((g[n] - base[n]).norm().item() / max(base[n].norm().item(), 1e-12), n) for n in g if "det_score.bias" not in n)[-3:]
    print("reserve %4d unfused %s  l2 cls0 %.5f total %.5f   worst rel. grad diff vs run 0:" % (reserve, fuse, l2["loss_ref_cls0"], sum(l2.values())), [(n.split(".")[-2] + "." + n.split(".")[-1], "%.1e" % d) for d, n in dev_])
