"""Counterpart of the reference's tools/train_net.py for the MI355X build: same command line
(--config-file <yaml> [KEY VALUE ...]), same yaml files, same flow (build model -> optimiser + WarmupMultiStepLR
-> load cfg.MODEL.WEIGHT if it is a local .pth -> train loop with periodic checkpoints in the reference's layout),
one process per GPU under torch.distributed.run (RCCL).

Batches come from the configured datasets (DATASETS.TRAIN + PROPOSAL_FILES.TRAIN under --data-dir, VOC devkit layout;
DataLoader workers decode and plan, the GPU preprocesses -- od_wscl_amd/data) or, with --synthetic, from
formula-generated images / proposals / image labels of the configured shape (no dataset needed).

    python tools/train_net.py --config-file /path/to/voc07_contra_db_b8_lr0.01_mcg.yaml --synthetic \\
        SOLVER.MAX_ITER 50 SOLVER.CHECKPOINT_PERIOD 25 OUTPUT_DIR /tmp/odw_run
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_loader(cfg, rank, device, size, proposals, max_iter, start_iter=0):
    from od_wscl_amd import synthetic
    from od_wscl_amd.structures import BoxList, to_image_list
    classes = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES
    div = cfg.DATALOADER.SIZE_DIVISIBILITY or 32
    for it in range(start_iter, max_iter):           # a resumed run continues the sequence where it stopped
        key = it * 1000 + rank                       # a different image per (iteration, rank)
        img = torch.from_numpy(synthetic.make_image(cfg.SEED, key, size, size))[:, :size, :size]
        boxes = torch.from_numpy(synthetic.make_proposals(cfg.SEED, key, proposals, size, size))
        labels = synthetic.make_labels(cfg.SEED, key, classes).tolist()
        t = BoxList(torch.zeros((len(labels), 4), device=device), (size, size), "xyxy")
        t.add_field("labels", torch.tensor(labels, device=device))
        t.add_field("labels_host", labels)
        yield to_image_list([img], div).to(device), [t], [BoxList(boxes.to(device), (size, size), "xyxy")]


def dataset_loader(cfg, device, world, start_iter, data_dir):
    """(images on the device, targets, proposals) from make_data_loader: the pending batch is materialised by the GPU
    preprocessing kernel in `.to(device)`; image labels stay on the host for the loss."""
    from od_wscl_amd.data import DatasetCatalog, make_data_loader
    if data_dir:
        DatasetCatalog.DATA_DIR = data_dir
    loader = make_data_loader(cfg, is_train=True, is_distributed=world > 1, start_iter=start_iter)
    for images, targets, rois, _ in loader:
        out_t = []
        for t in targets:
            t.add_field("labels_host", t.get_field("labels").tolist())
            out_t.append(t.to(device))
        yield images.to(device), out_t, [r.to(device) for r in rois]


def main():
    ap = argparse.ArgumentParser(description="OD-WSCL training on MI355X")
    ap.add_argument("--config-file", default="", metavar="FILE")
    ap.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    ap.add_argument("--synthetic", action="store_true", help="formula-generated batches instead of DATASETS.TRAIN")
    ap.add_argument("--data-dir", default="", help="root of the dataset catalog (default: ./datasets)")
    ap.add_argument("--size", type=int, default=600)
    ap.add_argument("--proposals", type=int, default=2000)
    ap.add_argument("--dtype", default="bf16x2f", choices=["bf16x2f", "bf16", "bf16x3", "bf16x2", "f32"],
                    help="arithmetic of the MFMA products (od_wscl_amd/precision.py); bf16x2f = forward at the parity bar, bf16 backward")
    ap.add_argument("--log-period", type=int, default=20)
    ap.add_argument("--allow-random-init", action="store_true",
                    help="train from the formula initialisation when MODEL.WEIGHT cannot be resolved locally")
    ap.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("tools/train_net.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(args.local_rank)
    device = torch.device("cuda", args.local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")

    from od_wscl_amd import engine
    from od_wscl_amd.config import cfg as base
    from od_wscl_amd.utils import checkpoint as ck
    from od_wscl_amd.utils.device_rand import DeviceRand
    cfg = base.clone()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts or [])
    if cfg.SEED < 0:
        cfg.merge_from_list(["SEED", 1234])
    if not args.synthetic and not cfg.DATASETS.TRAIN:
        raise SystemExit("DATASETS.TRAIN is empty: pass a config that names a dataset, or --synthetic")

    step, info = engine.build_training_step(cfg, device, dtype=args.dtype, world=world, seed=cfg.SEED)
    opt, model = step.optimizer, step.model
    start_iter = 0
    # resume order of the reference (utils/checkpoint.py:65-84): OUTPUT_DIR/last_checkpoint first, then MODEL.WEIGHT
    resume_from = ck.last_checkpoint(cfg.OUTPUT_DIR)
    weight = resume_from or cfg.MODEL.WEIGHT
    if weight and os.path.isfile(weight):           # catalog:// and http:// sources need the network: not here
        rest = ck.load_checkpoint(model, weight)
        if resume_from:                             # our own run: momenta, schedule position, iteration
            start_iter = ck.restore_training_state(opt, model, rest)
        else:                                       # pretrained weights: a fresh schedule (trainer.py / train_net.py:75-77)
            opt.sync_from_params(model)
        if rank == 0:
            print("loaded %s (%s, iteration %d)" % (weight, "resume" if resume_from else "weights only", start_iter), flush=True)
    elif weight and not (args.synthetic or args.allow_random_init):
        raise SystemExit("MODEL.WEIGHT %r is not a local file (catalog:// and http:// sources need the network).  Pass the "
                         "path of a pretrained .pth (VGG16: keys features.N / classifier.{1,4}, matched by suffix), or "
                         "--allow-random-init to train from the formula initialisation" % weight)
    elif weight and rank == 0:
        print("MODEL.WEIGHT %r is not a local file: training from the formula initialisation" % weight, flush=True)

    if cfg.SOLVER.ITER_SIZE > 1:            # reference tools/train_net.py:344-355: MAX_ITER counts optimiser steps
        cfg.SOLVER.MAX_ITER = cfg.SOLVER.MAX_ITER * cfg.SOLVER.ITER_SIZE
    max_iter = cfg.SOLVER.MAX_ITER
    period = getattr(cfg.SOLVER, "CHECKPOINT_PERIOD", 0)
    out_dir = cfg.OUTPUT_DIR
    if rank == 0 and out_dir:
        os.makedirs(out_dir, exist_ok=True)
    loader = synthetic_loader(cfg, rank, device, args.size, args.proposals, max_iter, start_iter) if args.synthetic else \
        dataset_loader(cfg, device, world, start_iter, args.data_dir)
    t0, seen = time.time(), 0
    for iteration, (images, targets, rois) in enumerate(loader, start_iter):
        iteration += 1                                             # engine/trainer.py:94
        if iteration > max_iter:
            break
        # engine/trainer.py:81-84 skips a batch that holds an image without labels (VOC images whose objects are all
        # "difficult"); every rank must take the same decision or the gradient all-reduce would hang
        empty = int(any(len(t) < 1 for t in targets))
        if world > 1:
            flag = torch.tensor([empty], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            empty = int(flag.item())
        if empty:
            continue
        rand = DeviceRand(cfg.SEED + rank, first_stream=(1 << 20) + (iteration << 12), device=device)
        losses, accs = step(images, targets, rois, rand, iteration=iteration)
        seen += sum(len(r) for r in rois)
        log_now = iteration % args.log_period == 0 or iteration == max_iter
        if log_now and world > 1:                                  # trainer.py:108-111: rank 0 logs the mean over ranks
            losses = engine.reduce_loss_dict(dict(losses), world)
        if rank == 0 and log_now:
            torch.cuda.synchronize()
            dt = time.time() - t0
            txt = "  ".join("%s: %.4f" % (k, float(v.detach())) for k, v in losses.items())
            print("iter: %d  lr: %.6f  %s  | %.0f proposals/s/GPU" % (
                iteration, cfg.SOLVER.BASE_LR * opt.lr_scale, txt, seen / dt), flush=True)
            t0, seen = time.time(), 0
        if rank == 0 and out_dir and period and (iteration % period == 0 or iteration == max_iter):
            path = os.path.join(out_dir, "model_%07d.pth" % iteration if iteration != max_iter else "model_final.pth")
            ck.save_checkpoint(model, path, optimizer=opt, iteration=iteration)
            ck.tag_last_checkpoint(out_dir, path)                               # utils/checkpoint.py:120-123
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
