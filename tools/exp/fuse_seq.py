import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from od_wscl_amd import engine, gemm
from od_wscl_amd.utils.device_rand import DeviceRand
dev = torch.device("cuda", 0)
os.environ["ODW_NO_TIMER"] = "1"
for seq in sys.argv[1:]:
    early, fuse = seq[0] == "E", seq[1] == "F"
    os.environ["ODW_NO_EARLY_BWD"] = "0" if early else "1"
    os.environ["ODW_NO_PRED_FUSE"] = "0" if fuse else "1"
    cfg = bench.build_cfg(21)
    step, _ = engine.build_training_step(cfg, dev, dtype="bf16", world=1, seed=cfg.SEED, backend="hip")
    images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 224, 150, 21, dev)
    out = []
    for it in range(2):
        l, _ = step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20) + (it << 12), device=dev))
        out.append(round(float(l["loss_ref_cls0"]), 6))
    torch.cuda.synchronize()
    print(seq, out, "reserve", gemm.WgradBatch.reserve, flush=True)
