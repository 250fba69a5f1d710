"""bench.py -- OD-WSCL proposal-feature hot path on MI355X: proposals/sec, forward+backward.

    python bench.py --gpus N --steps K --warmup W          (N > 1 and no WORLD_SIZE: launches its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input per rank:
VGG16-OICR backbone forward -> ROIPool over P precomputed proposals -> fc6/fc7 (clean + DropBlock
passes) -> Sim_Net -> MIST predictor -> OD-WSCL loss (IoU sampling, object discovery, SupCon,
3 refinement branches) -> backward -> gradient all-reduce (N>1) -> SGD step.
Workload at N=1 = BASELINE.json configs[1]: VGG16, 2000 proposals, batch 1, 600 px (padded to
608x608); at N>1 every rank gets its own image (weak scaling, configs[2] shape at N=8), or, with
--global-batch B, B/N images per rank (strong scaling; tools/train_net.py:286-294 of the reference is
the flow: one process per GPU, IMS_PER_BATCH split over the ranks).  Inputs are resident in HBM before
the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
DTYPE_NOTE = {
    "bf16x2f": "forward products: two bf16 planes per operand, 3 plane products, fp32 accumulate (meets the parity bar: "
               "losses <= 1e-3 rel, index selection bit-exact vs the fp32 reference); backward products: bf16, fp32 accumulate",
    "bf16x3": "every product: three bf16 planes per operand, 6 plane products (fp32-grade), fp32 accumulate",
    "bf16x2": "every product: two bf16 planes per operand, 3 plane products, fp32 accumulate",
    "bf16": "every product: bf16 operands, fp32 accumulate (does NOT meet the 1e-3 loss / exact-selection bar)",
}
# algorithmic FLOPs per proposal of the step as executed (row-sparse backward of the clean pass),
# VGG16 / VOC at 608^2, P = 2000: 1.36 GFLOP (1.90 with the reference's dense autograd backward)
ALGORITHMIC_GFLOP_PER_PROPOSAL = {"vgg16": 1.36}
# every precision mode runs on the bf16 matrix cores (bf16x3 / bf16x2 = 6 / 3 bf16 plane products per fp32-grade
# product, all of them counted as executed FLOPs): one dense peak
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "bf16x3": 2500.0, "bf16x2": 2500.0, "bf16x2f": 2500.0}
# The reference trains from ImageNet-pretrained VGG16 at lr 0.01 (configs/voc/*.yaml).  There are no
# checkpoints here: with random-init weights lr 0.01 diverges to NaN within 4 steps, so the bench
# keeps the identical work (same SGD update, momentum, weight decay) at a learning rate that stays finite.
BENCH_LR = 1e-5
ARCH_NAME = {"vgg16": "VGG16", "r50": "R-50-C5"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)          # ten rotations over the five warmed-up images: ~0.45 s of timed region
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--proposals", type=int, default=2000)
    ap.add_argument("--size", type=int, default=600)
    ap.add_argument("--classes", type=int, default=21)
    ap.add_argument("--dtype", default=os.environ.get("ODW_DTYPE", "bf16x2f"), choices=["bf16", "bf16x3", "bf16x2", "bf16x2f"],
                    help="arithmetic of the MFMA products (od_wscl_amd/precision.py): bf16x2f = the headline (forward "
                         "products on two bf16 planes per operand: losses within 1e-3 of the fp32 reference and every "
                         "selection identical, tests/test_e2e_gpu.py + tests/test_fullsize_gpu.py; backward products on "
                         "one plane); bf16x3 / bf16x2 = every product split; bf16 = single plane everywhere (does NOT "
                         "meet the parity bar: reported as `secondary`)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short single-plane bf16 run reported beside the headline (N = 1 only)")
    ap.add_argument("--no-microbench", action="store_true",
                    help="skip the live pairwise_sim measurement (roofline.kernels) after the timed region")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fixed global batch (strong scaling): B/N images per rank; 0 = one image per rank (weak)")
    ap.add_argument("--arch", default="vgg16", choices=["vgg16", "r50"],
                    help="vgg16 = the headline workload (BASELINE.json configs[1]); r50 = the R-50-C5 config "
                         "(configs/voc/voc07_r50_c5_*.yaml), a secondary line")
    ap.add_argument("--time-every", type=int, default=13,     # (coprime with the rotation: the timed steps fall on different images)
                    help="bracket the GEMM / conv launches of 1 timed step in N with HIP events (roofline object); such a step "
                         "carries ~220 event records and takes ~1.1 ms longer (`per_step_ms`), inside the timed region")
    ap.add_argument("--pooler", default="ROIPool", choices=["ROIPool", "ROIAlign"],
                    help="POOLER_METHOD; every shipped config of the reference uses ROIPool (the default)")
    ap.add_argument("--rotate", type=int, default=8,
                    help="the timed steps rotate over this many synthetic images per rank (image = rank + step mod n; 1-3 labels "
                         "each, `config.labels_per_image`): the loss's loop work is proportional to 3 x the image's positive "
                         "classes (weak_head/loss.py:281-345), and image 0 alone has ONE.  Capped at the warm-up step count, so "
                         "that every image of the timed region has been through the step (allocator, planner caches) before")
    ap.add_argument("--first-image", type=int, default=0,
                    help="index of the first synthetic image of rank 0's rotation (--first-image 2 --rotate 1: the 3-label image alone)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timeline", default="",
                    help="write a per-step HOST timeline of the timed region to this JSON file (od_wscl_amd/utils/step_trace.py: "
                         "ms blocked in the two device->host reads, ms of index assembly, allocator / planner / ring misses per "
                         "step, GPU time between the marks); the marks cost ~0.1 ms per step")
    ap.add_argument("--cpu-proposals", type=int, default=2000)
    # ---- data-parallel knobs (N > 1).  The defaults are what `bench.py --gpus N` runs: RCCL, fp32 gradients on the wire,
    # equal stream priorities
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the gradient exchange: nccl (= RCCL over xGMI, the measured "
                         "configuration) | gloo (plumbing tests on a box with fewer GPUs than ranks, see --oversubscribe)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="let ranks share GPUs (rank r on device r mod device_count): exercises the launcher, the process "
                         "group and the exchange on a 1-GPU box; the line is marked \"oversubscribed\" and is NOT a scaling number")
    ap.add_argument("--grad-exchange", default="fp32", choices=["fp32", "bf16"],
                    help="element type of the gradients on the wire (ODW.GRAD_EXCHANGE): fp32 = the reference's DDP; bf16 = "
                         "half the xGMI bytes, rounded once before the sum, fp32 again in the optimiser")
    ap.add_argument("--hp-stream", default="auto", choices=["auto", "0", "1"],
                    help="run the step on a high-priority HIP stream of its own: auto = off (round 6: the two stream hops per step "
                         "cost more than the priority buys, engine.py); 1 = on")
    return ap.parse_args()


def build_cfg(classes, arch="vgg16", pooler="ROIPool", grad_exchange="fp32"):
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
                         "nms", 0.1, "lmda", 0.03, "temp", 0.2, "SEED", 1234, "ODW.GRAD_EXCHANGE", grad_exchange])
    return cfg


def synthetic_batch(seed, rank, size, proposals, classes, device, n_images=1):
    """`n_images` images of this rank (image index = rank * n_images + k: no two ranks share one)."""
    from od_wscl_amd import synthetic
    from od_wscl_amd.structures import BoxList, to_image_list
    imgs, rois, targets = [], [], []
    for k in range(n_images):
        idx = rank * n_images + k
        img = torch.from_numpy(synthetic.make_image(seed, idx, size, size))
        boxes = torch.from_numpy(synthetic.make_proposals(seed, idx, proposals, size, size))
        labels = torch.from_numpy(synthetic.make_labels(seed, idx, classes))
        imgs.append(img[:, :size, :size])
        rois.append(BoxList(boxes.to(device), (size, size), "xyxy"))
        t = BoxList(torch.zeros((len(labels), 4), device=device), (size, size), "xyxy")
        t.add_field("labels", labels.to(device))
        t.add_field("labels_host", labels.tolist())     # the data loader has the image labels on the host anyway
        targets.append(t)
    images = to_image_list(imgs, 32).to(device)
    return images, targets, rois


def cpu_baseline(args, seed):
    """The CPU oracle ("port" of the reference path, validated against the reference's golden vectors) timed on this
    host: the same synthetic step, forward + backward, at 8 / 32 / all physical cores."""
    from od_wscl_amd import synthetic
    from oracle import hotpath_ref as H
    p = args.cpu_proposals
    sd = H.make_state(1, args.classes, arch=args.arch)
    # three of the images the GPU line rotates over: 1, 2 and 3 labels (mean 2.0 = that of the default rotation 0..4)
    sample = [(torch.from_numpy(synthetic.make_image(seed, i, args.size, args.size))[None],
               [torch.from_numpy(synthetic.make_proposals(seed, i, p, args.size, args.size))],
               [torch.from_numpy(synthetic.make_labels(seed, i, args.classes))]) for i in range(3)]
    cfg = dict(nms=0.1, lmda=0.03, thres=0.5, temp=0.2, pooler="ROIPool", arch=args.arch,
               scale=0.125 if args.arch == "vgg16" else 0.0625)
    turn = [0]

    def one():
        img, boxes, labels = sample[turn[0] % len(sample)]
        turn[0] += 1
        for v in sd.values():
            if getattr(v, "grad", None) is not None:
                v.grad = None
        t0 = time.time()
        losses, _ = H.forward(img, boxes, labels, sd, H.Rand(seed), cfg)
        sum(losses.values()).backward()
        return time.time() - t0

    # thread counts: 8 and 32 (median of 3 after a warm-up each), then every physical core once warmed (2 steps, the
    # better one) unless a step there would blow the time budget -- on a 2 x 64-core host the small ops of the loss
    # oversubscribe and the step gets SLOWER with the core count (measured: 256 threads 79 s, 8 threads 6 s)
    logical = os.cpu_count() or 1
    physical = max(1, logical // 2)
    runs = {}
    for cores in (8, 32):
        if cores > logical or cores in runs:
            continue
        torch.set_num_threads(cores)
        turn[0] = 2
        one()                                                   # warm-up (thread pool, allocator, first-touch pages)
        runs[cores] = sum(one() for _ in range(3)) / 3.0        # mean over images 0, 1, 2 (1, 2 and 3 labels)
    if physical not in runs and physical > 32 and min(runs.values()) < 8.0:
        torch.set_num_threads(physical)
        first = one()
        runs[physical] = min(first, one()) if first < 20.0 else first
    best = min(runs, key=lambda c: runs[c])
    dt = runs[best]
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(p / dt, 2), "unit": "proposals/s", "cores": best, "kind": "port",
            "sample": "mean of 3 steps (synthetic images 0, 1, 2: 1, 2 and 3 labels) after 1 warm-up, fwd+bwd (no optimizer), "
                      "%s %dpx, %d of the %d proposals, torch-CPU fp32 oracle (C ROIPool single-threaded); `value` = the "
                      "fastest thread count tried"
                      % (ARCH_NAME[args.arch], args.size, p, args.proposals),
            "seconds": round(dt, 2), "cpu_model": model, "host_logical_cpus": logical,
            "by_threads": {str(c): round(p / t, 2) for c, t in sorted(runs.items())}}


def pairwise_sim_live(device, sizes=(2000, 4000, 8000), iters=30):
    """The drop-in P x P similarity kernel (reference: roi_heads/weak_head/loss.py:319, sim_mat = E E^T) measured live
    with HIP events on the launch stream: algorithmic bytes 4 P^2 + 512 P over the average launch time,
    against the 8 TB/s HBM roofline (north_star: >= 60 % at P = 4000).  (The training step itself computes only the
    similarity ROWS object discovery reads, so this operator is measured on its own.)"""
    from od_wscl_amd import _lib as L
    lib = L.lib()
    out = {}
    for p in sizes:
        g = torch.Generator(device=device).manual_seed(p)
        e = torch.nn.functional.normalize(torch.randn(p, 128, device=device, generator=g), dim=1).contiguous()
        s_mat = torch.empty(p, p, device=device)
        wsb = lib.odw_pairwise_sim_workspace(p, 128)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=device)

        def launch():
            L.check(lib.odw_pairwise_sim_ws(L.ptr(e), p, 128, L.ptr(s_mat), L.ptr(ws), wsb, L.stream()), "pairwise_sim")
        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        # `iters` launches captured in ONE HIP graph and replayed: the events then bracket back-to-back kernels, not the
        # host's launch calls (a 13 us kernel launched through ctypes is host-bound)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(iters):
                launch()
        graph.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        graph.replay()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / iters * 1e3
        del graph
        nbytes = 4.0 * p * p + 512.0 * p
        out["pairwise_sim P=%d" % p] = {"bound": "hbm", "avg_launch_us": round(us, 2), "algorithmic_bytes": int(nbytes),
                                        "achieved_GBps": round(nbytes / us / 1e3, 1), "peak_GBps": HBM_PEAK_GBPS,
                                        "frac": round(nbytes / (us * 1e-6) / (HBM_PEAK_GBPS * 1e9), 4)}
    return out


def collective_report(exch, args, world, steps, shared):
    """The `collective` object of the line: what carried the gradient exchange and how much of it the step WAITED for.
    exposed_ms_per_step = time the step's own stream spent blocked in GradExchange.finish (HIP events on that stream,
    around the waits on the outstanding all-reduces: tools/train_net.py:50-55 of the reference is DDP's bucketed
    all-reduce); side_stream_wait_ms = the same for the optimiser's side stream, where the head's exchange is waited for
    under the body's backward (hidden unless it outlasts the backward)."""
    if world <= 1:
        return {"backend": None, "ranks": 1, "wire_dtype": None, "exposed_ms_per_step": 0.0}
    tot = {"main": 0.0, "side": 0.0}
    for tag, e0, e1 in exch.marks:
        tot[tag] += e0.elapsed_time(e1)
    nbytes = exch.flat.numel() * (2 if exch.dtype == "bf16" else 4)
    return {"backend": "rccl" if args.backend == "nccl" else args.backend, "ranks": world, "wire_dtype": exch.dtype,
            "bytes_per_step": int(nbytes), "exposed_ms_per_step": round(tot["main"] / steps, 4),
            "side_stream_wait_ms_per_step": round(tot["side"] / steps, 4),
            "hp_stream": os.environ.get("ODW_HP_STREAM", "auto"), "devices_shared": bool(shared)}


def launch_ranks(n, oversubscribe=False):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU over RCCL, the
    reference's torch.distributed.launch flow, tools/train_net.py:286-294) and relay rank 0's JSON line."""
    have = torch.cuda.device_count()
    if have < n and not oversubscribe:
        raise RuntimeError("bench.py --gpus %d: only %d GPU(s) visible on this node -- refusing to report an "
                           "n_gpus=%d number from fewer devices (--oversubscribe --backend gloo runs the plumbing "
                           "with shared devices)" % (n, have, n))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.oversubscribe and args.backend == "nccl" and args.gpus > torch.cuda.device_count():
        raise RuntimeError("bench.py: RCCL refuses two ranks on one device -- use --backend gloo with --oversubscribe")
    if args.hp_stream != "auto":
        os.environ["ODW_HP_STREAM"] = args.hp_stream
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, args.oversubscribe))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise RuntimeError("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if args.global_batch and args.global_batch % world:
        raise RuntimeError("bench.py: --global-batch %d is not a multiple of %d ranks" % (args.global_batch, world))
    ipr = args.global_batch // world if args.global_batch else 1            # images per rank
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and not args.oversubscribe:
        raise RuntimeError("bench.py: rank %d has no GPU of its own (%d visible)" % (local_rank, n_dev))
    shared = args.oversubscribe and world > n_dev
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, init_method="env://")   # nccl == RCCL on ROCm
        assert dist.get_world_size() == args.gpus
    from od_wscl_amd import _lib
    _lib.lib()                                                            # fail loudly if the .so is missing
    from od_wscl_amd import engine
    from od_wscl_amd.utils.device_rand import DeviceRand

    cfg = build_cfg(args.classes, args.arch, args.pooler, args.grad_exchange)
    seed = cfg.SEED
    # The timed steps rotate over `n_rot` batches of this rank (batch j = images (rank + j) * ipr ...): the synthetic images
    # carry 1-3 labels and the loss's loops run 3 x n_pos chains per image (weak_head/loss.py:281-345) -- image 0 alone (one
    # label, loss_sim == 0) is the lightest case the workload allows.  Never more batches than warm-up steps: every batch of
    # the timed region has been through the step once before the clock starts.
    n_rot = max(1, min(args.rotate, args.warmup if args.warmup > 0 else 1))
    batches = [synthetic_batch(seed, args.first_image + rank + j, args.size, args.proposals, args.classes, device, n_images=ipr)
               for j in range(n_rot)]
    images, targets, rois = batches[0]
    labels_per_image = [[len(t.get_field("labels_host")) for t in b[1]] for b in batches]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(dtype, steps, warmup):
        """warmup untimed steps, then exactly `steps` timed ones between two barriers; returns (seconds = max over ranks,
        per-step GPU milliseconds from HIP events on the launch stream, info of the step, roofline object of rank 0)."""
        step_fn, info = engine.build_training_step(cfg, device, dtype=dtype, world=world, seed=seed)
        exch = step_fn.optimizer.exchange
        for it in range(warmup):
            bi, bt, br = batches[it % n_rot]
            step_fn(bi, bt, br, DeviceRand(seed + rank, first_stream=(1 << 20) + (it << 12), device=device))
        n_timed = len([it for it in range(steps) if it % args.time_every == 0])
        engine.kernel_timer.reset(prealloc=2 * 160 * n_timed if engine.kernel_timer.enabled else 0)
        engine.kernel_timer.timed_steps = n_timed
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        exch.measure, exch.marks = world > 1, []        # event pairs around GradExchange.finish (the waits on the collectives)
        # the model, the optimiser and the captured graphs are long-lived: keep the cyclic collector from walking them in the
        # middle of a step (a full collection over ~1e6 objects showed up as a single 17 ms step in `per_step_ms`)
        gc.collect()
        gc.freeze()
        barrier()
        allocs0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
        t0 = time.perf_counter()
        trace = engine.step_trace if (args.timeline and dtype == args.dtype) else None
        if trace is not None:
            trace.enabled, trace.gpu_events, trace.steps = True, True, []
        host_s = 0.0
        for it in range(steps):
            engine.kernel_timer.active = it % args.time_every == 0
            marks[it].record()
            bi, bt, br = batches[(warmup + it) % n_rot]
            if trace is not None:
                trace.begin(image=args.first_image + rank + (warmup + it) % n_rot, labels=labels_per_image[(warmup + it) % n_rot],
                            event_step=bool(engine.kernel_timer.active))
            th = time.perf_counter()
            step_fn(bi, bt, br, DeviceRand(seed + rank, first_stream=(1 << 20) + ((warmup + it) << 12), device=device))
            host_s += time.perf_counter() - th
            if trace is not None:
                trace.end()
        marks[steps].record()
        barrier()
        dt = time.perf_counter() - t0
        engine.kernel_timer.active = True
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        gc.unfreeze()
        info = dict(info, device_allocs=int(torch.cuda.memory_stats(device).get("num_device_alloc", 0) - allocs0),
                    host_ms_per_step=host_s / steps * 1e3,
                    step_images=[(warmup + it) % n_rot for it in range(steps)])
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        if trace is not None:
            trace.enabled = False
            if rank == 0:
                rep = trace.report(per_step)
                rep["cmd"] = " ".join(sys.argv)
                rep["wall_ms_per_step"] = round(dt / steps * 1e3, 3)
                with open(args.timeline, "w") as f:
                    json.dump(rep, f, indent=1)
        exch.measure = False
        info = dict(info, collective=collective_report(exch, args, world, steps, shared))
        roof = engine.kernel_timer.roofline(dtype, MFMA_PEAK_TFLOPS, HBM_PEAK_GBPS) if rank == 0 else None
        hbm = engine.kernel_timer.hbm_entries(HBM_PEAK_GBPS) if rank == 0 else None
        flops_step = engine.kernel_timer.flops_per_timed_step() if rank == 0 else None
        del step_fn
        torch.cuda.empty_cache()
        return dt, per_step, info, roof, hbm, flops_step

    dt, per_step, info, roof, hbm, flops_step = run(args.dtype, args.steps, args.warmup)
    ms = dt / args.steps * 1e3
    value = world * ipr * args.proposals * args.steps / dt
    med = float(np.median(per_step))

    by_labels = {}
    for it, ms_it in zip(info.get("step_images", []), per_step):
        by_labels.setdefault(str(sum(labels_per_image[it])), []).append(ms_it)
    by_labels = {k: round(float(np.mean(v)), 3) for k, v in sorted(by_labels.items())}     # GPU ms of a step by its batch's label count
    if rank == 0:
        if roof is not None:
            # memory-side bytes of the dominant symbol's heaviest launch shape from the committed PMC passes (rocprofv3 cannot
            # run inside the timed region): FETCH_SIZE x 2 (gfx950 under-count of 16-B/lane reads) + WRITE_SIZE, per launch.
            # Keyed on (kernel, dtype, launch shape): a run at another size reports null instead of another shape's bytes.
            for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
                tpath = os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")
                if os.path.exists(tpath):
                    t = json.load(open(tpath))
                    entries = t if isinstance(t, list) else [t]
                    hit = [e for e in entries if e.get("kernel") == roof["kernel"] and e.get("dtype", "bf16") == args.dtype
                           and e.get("shape") == roof.get("top_shape")]
                    if hit:
                        roof["traffic"] = hit[0]["traffic_bytes"]
                        roof["traffic_of"] = "%s [%s]; algorithmic %d B; %s" % (hit[0]["launch"], hit[0]["shape"],
                                                                               hit[0]["algorithmic_bytes"], hit[0]["source"])
                        break
            roof["note"] = ("dominant = the kernel SYMBOL with the largest total time over the timed steps, split-K launches "
                            "included; frac = FLOPs of the reference's fp32 arithmetic for what its launches deliver / their time / "
                            "peak; frac_issued counts every bf16 plane product issued (3 per fp32-grade forward product). "
                            "gemm_nt_cm_kernel<true, 1> delivers TWO fc6 evaluations (clean + DropBlock) per sweep.")
            roof["timed_steps"] = "HIP events around every GEMM/conv launch of 1 timed step in %d" % args.time_every
            # ---- the whole step against the MFMA roofline: algorithmic work (1.36 GFLOP per proposal as
            # executed, single-precision semantics) and the MFMA work actually issued (every plane product counted)
            alg = ALGORITHMIC_GFLOP_PER_PROPOSAL.get(args.arch) if (args.proposals, args.size, args.classes) == (2000, 600, 21) else None
            e2e = {"peak_TFLOPs": MFMA_PEAK_TFLOPS[args.dtype]}
            if alg is not None:
                e2e["algorithmic_GFLOP_per_proposal"] = alg
                e2e["algorithmic_TFLOPs"] = round(value / world * alg * 1e9 / 1e12, 2)
                e2e["frac_algorithmic"] = round(value / world * alg * 1e9 / (MFMA_PEAK_TFLOPS[args.dtype] * 1e12), 4)
            if flops_step:
                e2e["issued_TFLOP_per_step"] = round(flops_step / 1e12, 3)
                e2e["issued_TFLOPs"] = round(flops_step / (med * 1e-3) / 1e12, 2)
                e2e["frac_issued"] = round(flops_step / (med * 1e-3) / (MFMA_PEAK_TFLOPS[args.dtype] * 1e12), 4)
                e2e["note"] = "issued = sum of 2MNK over every GEMM / convolution launch of a timed step (plane products counted)"
            roof["end_to_end"] = e2e
            kernels = dict(hbm or {})
            if not args.no_microbench and world == 1:
                kernels.update(pairwise_sim_live(device))
            roof["kernels"] = kernels
        out = {
            "metric": "proposals/sec fwd+bwd (%s, %d proposals, %dpx)" % (ARCH_NAME[args.arch], args.proposals, args.size),
            "value": round(value, 1), "unit": "proposals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None, "dtype": args.dtype, "dtype_note": DTYPE_NOTE[args.dtype], "data": "synthetic",
            "median_ms_per_step": round(med, 3),
            "value_at_median_step": round(world * ipr * args.proposals / (med * 1e-3), 1),
            "per_step_ms": [round(float(v), 2) for v in per_step],      # rank 0's HIP-event time of each timed step
            "device_allocs_in_timed_region": info.get("device_allocs"),   # hipMalloc calls of the caching allocator (a spike in per_step_ms)
            # host time spent INSIDE the step call per step (launching; the call returns with GPU work queued): well below
            # ms_per_step = the launching thread runs ahead of the GPU and a host hiccup does not reach the step time.  The loss
            # reads nothing back (csrc/loss_lists.hip); the one wait left is the back-pressure of the index-table ring (8 steps).
            "host_ms_per_step": round(info.get("host_ms_per_step", 0.0), 3),
            "host_reads_per_step": 0 if (args.dtype == "bf16x2f" and os.environ.get("ODW_HOST_LISTS") != "1") else 2,
            "ms_per_step_by_labels": by_labels,
            "config": {"workload": "%s + %d MCG-like proposals, batch %d/GPU, %dpx (padded %d), %s 7x7, "
                                   "OD-WSCL loss (CONTRA), SGD step, %d classes; timed steps rotate over %d synthetic image(s) "
                                   "with labels_per_image %s (mean %.2f)"
                                   % ("VGG16-OICR" if args.arch == "vgg16" else "R-50-C5", args.proposals, ipr, args.size,
                                      images.tensors.shape[-1], args.pooler, args.classes, n_rot,
                                      [n for b in labels_per_image for n in b],
                                      float(np.mean([n for b in labels_per_image for n in b]))),
                       "labels_per_image": labels_per_image, "rotate": n_rot,
                       "global_batch": world * ipr, "parallelism": "dp%d" % world, "world_size": world, "lr": BENCH_LR, "gemm_backend": info["gemm_backend"],
                       "conv_backend": info["conv_backend"], "optimizer": info["optimizer"]},
            "per_gpu": round(value / world, 1),
            "collective": info["collective"],
            "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("ODW_") or k == "GPU_MAX_HW_QUEUES"},
            "roofline": roof,
        }
    if world == 1 and not args.no_secondary and args.dtype != "bf16":
        # the single-plane mode, for reference only: same workload, a short run (it misses the parity bar)
        n = max(5, args.steps // 2)
        sdt, sper, _, _, _, _ = run("bf16", n, max(3, n_rot))
        if rank == 0:
            out["secondary"] = {"bf16": {"value": round(args.proposals * ipr * n / sdt, 1), "unit": "proposals/s",
                                         "ms_per_step": round(sdt / n * 1e3, 3), "median_ms_per_step": round(float(np.median(sper)), 3),
                                         "steps": n, "note": DTYPE_NOTE["bf16"]}}
        if args.dtype != "bf16x3":
            # the fp32-GRADE mode (the reference computes in fp32, config/defaults.py:559): every product, forward and
            # backward, on three bf16 planes per operand = six plane products.  A short run of the same workload.
            n3 = max(5, args.steps // 4)
            sdt, sper, _, _, _, _ = run("bf16x3", n3, max(2, n_rot))
            if rank == 0:
                out["secondary"]["bf16x3"] = {"value": round(args.proposals * ipr * n3 / sdt, 1), "unit": "proposals/s",
                                              "ms_per_step": round(sdt / n3 * 1e3, 3),
                                              "median_ms_per_step": round(float(np.median(sper)), 3), "steps": n3,
                                              "note": DTYPE_NOTE["bf16x3"]}
    if rank == 0:
        if shared:
            out["oversubscribed"] = "%d ranks on %d device(s): plumbing run, NOT a scaling measurement" % (world, n_dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
