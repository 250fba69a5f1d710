"""bench.py -- OD-WSCL proposal-feature hot path on MI355X: proposals/sec, forward+backward.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input per rank:
VGG16-OICR backbone forward -> ROIPool over P precomputed proposals -> fc6/fc7 (clean + DropBlock
passes) -> Sim_Net -> MIST predictor -> OD-WSCL loss (IoU sampling, object discovery, SupCon,
3 refinement branches) -> backward -> gradient all-reduce (N>1) -> SGD step.
Workload at N=1 = BASELINE.json configs[1]: VGG16, 2000 proposals, batch 1, 600 px (padded to
608x608); at N>1 every rank gets its own image (weak scaling, configs[2] shape).  Inputs are
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}
# The reference trains from ImageNet-pretrained VGG16 at lr 0.01 (configs/voc/*.yaml).  There are no
# checkpoints here: with random-init weights lr 0.01 diverges to NaN within 4 steps, so the bench
# keeps the identical work (same SGD update, momentum, weight decay) at a learning rate that stays finite.
BENCH_LR = 1e-5
ARCH_NAME = {"vgg16": "VGG16", "r50": "R-50-C5"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--proposals", type=int, default=2000)
    ap.add_argument("--size", type=int, default=600)
    ap.add_argument("--classes", type=int, default=21)
    ap.add_argument("--dtype", default=os.environ.get("ODW_DTYPE", "bf16"), choices=["bf16", "f32"])
    ap.add_argument("--backend", default=os.environ.get("ODW_BACKEND", "hip"), choices=["hip", "torch"],
                    help="hip = hand-written gfx950 kernels for the ROI head (default); torch = library comparison")
    ap.add_argument("--arch", default="vgg16", choices=["vgg16", "r50"],
                    help="vgg16 = the headline workload (BASELINE.json configs[1]); r50 = the R-50-C5 config "
                         "(configs/voc/voc07_r50_c5_*.yaml), a secondary line")
    ap.add_argument("--time-every", type=int, default=4,
                    help="bracket the GEMM / conv launches of 1 timed step in N with HIP events (roofline object)")
    ap.add_argument("--pooler", default="ROIPool", choices=["ROIPool", "ROIAlign"],
                    help="POOLER_METHOD; every shipped config of the reference uses ROIPool (the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-proposals", type=int, default=2000)
    return ap.parse_args()


def build_cfg(classes, arch="vgg16", pooler="ROIPool"):
    from od_wscl_amd.config import make_defaults
    cfg = make_defaults()
    # == configs/voc/voc07_contra_db_b8_lr0.01_mcg.yaml of the reference (voc07_r50_c5_contra_db_b8_lr0.02_ss.yaml for r50)
    body = (["MODEL.BACKBONE.CONV_BODY", "VGG16-OICR", "MODEL.ROI_BOX_HEAD.POOLER_SCALES", (0.125,),
             "MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR", "VGG16.roi_head"] if arch == "vgg16" else
            ["MODEL.BACKBONE.CONV_BODY", "R-50-C5", "MODEL.ROI_BOX_HEAD.POOLER_SCALES", (0.0625,),
             "MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR", "ResNet50Conv5ROIFeatureExtractor"])
    cfg.merge_from_list(body + ["MODEL.WSOD_ON", True, "MODEL.FASTER_RCNN", False,
                         "MODEL.ROI_BOX_HEAD.NUM_CLASSES", classes, "MODEL.ROI_BOX_HEAD.POOLER_METHOD", pooler,
                         "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", 7,
                         "MODEL.ROI_WEAK_HEAD.REGRESS_ON", True, "DB.METHOD", "dropblock", "SOLVER.CONTRA", True,
                         "SOLVER.BASE_LR", BENCH_LR, "SOLVER.WEIGHT_DECAY", 0.0001, "SOLVER.IMS_PER_BATCH", 8,
                         "nms", 0.1, "lmda", 0.03, "temp", 0.2, "SEED", 1234])
    return cfg


def make_optimizer(cfg, model):
    """solver/build.py:10-24: per-parameter groups, bias lr x2 and no weight decay."""
    params = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
        if "bias" in key:
            lr, wd = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR, cfg.SOLVER.WEIGHT_DECAY_BIAS
        params.append({"params": [value], "lr": lr, "weight_decay": wd})
    return torch.optim.SGD(params, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)


def synthetic_batch(seed, rank, size, proposals, classes, device):
    from od_wscl_amd import synthetic
    from od_wscl_amd.structures import BoxList, to_image_list
    img = torch.from_numpy(synthetic.make_image(seed, rank, size, size))
    boxes = torch.from_numpy(synthetic.make_proposals(seed, rank, proposals, size, size))
    labels = torch.from_numpy(synthetic.make_labels(seed, rank, classes))
    images = to_image_list([img[:, :size, :size]], 32).to(device)
    rois = [BoxList(boxes.to(device), (size, size), "xyxy")]
    t = BoxList(torch.zeros((len(labels), 4), device=device), (size, size), "xyxy")
    t.add_field("labels", labels.to(device))
    t.add_field("labels_host", labels.tolist())     # the data loader has the image labels on the host anyway
    return images, [t], rois


def cpu_baseline(args, seed):
    """The CPU oracle ("port" of the reference path, validated against the reference's golden
    vectors) timed on this host: one forward+backward on a bounded sample of the same workload."""
    from od_wscl_amd import synthetic
    from oracle import hotpath_ref as H
    p = args.cpu_proposals
    cores = min(os.cpu_count() or 1, 32)     # beyond ~32 threads torch-CPU GEMMs of this size stop scaling
    torch.set_num_threads(cores)
    sd = H.make_state(1, args.classes, arch=args.arch)
    img = torch.from_numpy(synthetic.make_image(seed, 0, args.size, args.size))[None]
    boxes = [torch.from_numpy(synthetic.make_proposals(seed, 0, p, args.size, args.size))]
    labels = [torch.from_numpy(synthetic.make_labels(seed, 0, args.classes))]
    cfg = dict(nms=0.1, lmda=0.03, thres=0.5, temp=0.2, pooler="ROIPool", arch=args.arch,
               scale=0.125 if args.arch == "vgg16" else 0.0625)
    t0 = time.time()
    losses, _ = H.forward(img, boxes, labels, sd, H.Rand(seed), cfg)
    sum(losses.values()).backward()
    dt = time.time() - t0
    return {"value": round(p / dt, 2), "unit": "proposals/s", "cores": cores, "kind": "port",
            "sample": "1 step fwd+bwd (no optimizer), %s %dpx, %d of the %d proposals, torch-CPU fp32 oracle (C ROIPool single-threaded), %d threads"
                      % (ARCH_NAME[args.arch], args.size, p, args.proposals, cores), "seconds": round(dt, 2)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")     # nccl == RCCL on ROCm
    from od_wscl_amd import _lib
    _lib.lib()                                                            # fail loudly if the .so is missing
    from od_wscl_amd import engine
    from od_wscl_amd.utils.device_rand import DeviceRand

    cfg = build_cfg(args.classes, args.arch, args.pooler)
    seed = cfg.SEED
    step_fn, info = engine.build_training_step(cfg, device, dtype=args.dtype, world=world, seed=seed,
                                                 backend=args.backend)
    images, targets, rois = synthetic_batch(seed, rank, args.size, args.proposals, args.classes, device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        step_fn(images, targets, rois, DeviceRand(seed + rank, first_stream=(1 << 20) + (it << 12), device=device))
    engine.kernel_timer.reset()
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        engine.kernel_timer.active = it % args.time_every == 0
        step_fn(images, targets, rois,
                DeviceRand(seed + rank, first_stream=(1 << 20) + ((args.warmup + it) << 12), device=device))
    barrier()
    dt = time.perf_counter() - t0
    engine.kernel_timer.active = True
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = world * args.proposals * args.steps / dt

    if rank == 0:
        roof = engine.kernel_timer.roofline(args.dtype, MFMA_PEAK_TFLOPS, HBM_PEAK_GBPS)
        if roof is not None:
            # memory-side bytes of the dominant launch from the committed PMC passes (rocprofv3 cannot run inside the
            # timed region): FETCH_SIZE x 2 (gfx950 under-count of 16-B/lane reads) + WRITE_SIZE, per launch
            tpath = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
            if os.path.exists(tpath):
                t = json.load(open(tpath))
                if t.get("kernel") == roof["kernel"]:
                    roof["traffic"] = t["traffic_bytes"]
                    roof["traffic_of"] = "%s; algorithmic %d B; %s" % (t["launch"], t["algorithmic_bytes"], t["source"])
            roof["timed_steps"] = "HIP events around every GEMM/conv launch of 1 timed step in %d" % args.time_every
        out = {
            "metric": "proposals/sec fwd+bwd (%s, %d proposals, %dpx)" % (ARCH_NAME[args.arch], args.proposals, args.size),
            "value": round(value, 1), "unit": "proposals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s + %d MCG-like proposals, batch 1/GPU, %dpx (padded %d), %s 7x7, "
                                   "OD-WSCL loss (CONTRA), SGD step, %d classes"
                                   % ("VGG16-OICR" if args.arch == "vgg16" else "R-50-C5", args.proposals, args.size, images.tensors.shape[-1],
                                      args.pooler, args.classes),
                       "global_batch": world, "parallelism": "dp%d" % world, "lr": BENCH_LR, "gemm_backend": info["gemm_backend"],
                       "conv_backend": info["conv_backend"], "optimizer": info["optimizer"]},
            "per_gpu": round(value / world, 1),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
