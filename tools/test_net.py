"""Counterpart of the reference's tools/test_net.py: evaluate a checkpoint on DATASETS.TEST (single scale or the
test-time augmentation of TEST.BBOX_AUG), one process per GPU, VOC mAP on rank 0.

    python tools/test_net.py --config-file <yaml> --data-dir <root with voc/VOC2007 + proposal files> \\
        MODEL.WEIGHT /path/model_final.pth OUTPUT_DIR /tmp/odw_eval
"""
import argparse
import logging
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_eval_model(cfg, device, dtype="bf16"):
    """Eval-mode detector on the gfx950 kernels; dtype = od_wscl_amd.precision mode ("bf16" | "bf16x3" | "bf16x2")."""
    from od_wscl_amd import precision
    from od_wscl_amd.modeling.detector import build_detection_model
    precision.set_precision({"fp32": "bf16x3", "f32": "bf16x3"}.get(dtype, dtype))
    model = build_detection_model(cfg).to(device)
    model.eval()
    model.hip_body()
    return model


def main():
    ap = argparse.ArgumentParser(description="OD-WSCL evaluation on MI355X")
    ap.add_argument("--config-file", default="", metavar="FILE")
    ap.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    ap.add_argument("--data-dir", default="")
    ap.add_argument("--allow-random-init", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "bf16x3", "bf16x2", "f32"])
    ap.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise RuntimeError("tools/test_net.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(args.local_rank)
    device = torch.device("cuda", args.local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")
    logging.basicConfig(level=logging.INFO)

    from od_wscl_amd import inference
    from od_wscl_amd.config import cfg as base
    from od_wscl_amd.data import DatasetCatalog, make_data_loader
    from od_wscl_amd.utils import checkpoint as ck
    cfg = base.clone()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts or [])
    if args.data_dir:
        DatasetCatalog.DATA_DIR = args.data_dir
    model = build_eval_model(cfg, device, args.dtype)
    weight = ck.last_checkpoint(cfg.OUTPUT_DIR) if not (cfg.MODEL.WEIGHT and os.path.isfile(cfg.MODEL.WEIGHT)) else cfg.MODEL.WEIGHT
    if weight and os.path.isfile(weight):
        ck.load_checkpoint(model, weight)
    elif not args.allow_random_init:
        raise SystemExit("MODEL.WEIGHT %r is not a local file and OUTPUT_DIR holds no checkpoint: an evaluation of "
                         "randomly initialised weights is meaningless (pass --allow-random-init to run it anyway)"
                         % cfg.MODEL.WEIGHT)
    for name in cfg.DATASETS.TEST:
        if "coco" in name:
            raise SystemExit("dataset %r: the COCO bbox metric (pycocotools' COCOeval) is outside this build -- only the "
                             "VOC07 metric is implemented (od_wscl_amd/data/evaluation.py)" % name)
    loaders = make_data_loader(cfg, is_train=False, is_distributed=world > 1)
    for name, loader in zip(cfg.DATASETS.TEST, loaders):
        out = os.path.join(cfg.OUTPUT_DIR, "inference", name) if cfg.OUTPUT_DIR else None
        if out and (not dist.is_initialized() or dist.get_rank() == 0):
            os.makedirs(out, exist_ok=True)
        result = inference.inference(model, loader, name, cfg, device=device, output_folder=out)
        if result is not None:
            print("%s: mAP %.4f" % (name, result["map"]))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
