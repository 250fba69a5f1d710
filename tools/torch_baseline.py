"""Library comparison step (NOT part of the product): the same model with every Linear on F.linear (hipBLASLt), the
body on torch's own modules (MIOpen) and torch.optim.SGD, timed like bench.py.  The product package has no such
path any more -- this script monkeypatches od_wscl_amd.gemm.fused_linear and GeneralizedRCNN.hip_body from outside.

    python tools/torch_baseline.py [--dtype f32|bf16] [--steps 20] [--warmup 5]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def torch_fused_linear(x, weight, bias, shadow, relu=False, drop_p=0.0, segs=None, out_f32=False, tag=None, grad_rows=None,
                       row_ids=None):
    from od_wscl_amd.utils.device_rand import dropout_with_segments
    y = F.linear(x.to(weight.dtype) if x.dtype != weight.dtype and not torch.is_autocast_enabled() else x, weight, bias)
    if relu:
        y = torch.relu(y)
    if drop_p > 0:
        assert row_ids is None, "the library path has no row-sparse re-evaluation"
        y = dropout_with_segments(y, drop_p, segs)
    return y.float() if out_f32 else y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--proposals", type=int, default=2000)
    ap.add_argument("--size", type=int, default=600)
    args = ap.parse_args()
    import bench
    from od_wscl_amd import engine, gemm, precision
    from od_wscl_amd.modeling.detector import build_detection_model
    from od_wscl_amd.modeling.detector.generalized_rcnn import GeneralizedRCNN
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    precision.set_precision("bf16x3")          # fp32 activations between ops; no HIP GEMM is called below
    gemm.fused_linear = torch_fused_linear
    import od_wscl_amd.modeling.roi_heads.weak_head.roi_weak_predictors as rp
    rp.gemm = gemm
    GeneralizedRCNN.hip_body = lambda self: self.backbone
    cfg = bench.build_cfg(21)
    model = build_detection_model(cfg).to(dev)
    engine.load_formula_weights(model, 1)
    model.train()
    groups = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
        if "bias" in key:
            lr, wd = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR, cfg.SOLVER.WEIGHT_DECAY_BIAS
        groups.append({"params": [value], "lr": lr, "weight_decay": wd})
    opt = torch.optim.SGD(groups, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)
    images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, args.size, args.proposals, 21, dev)
    amp = args.dtype == "bf16"

    def step(it):
        rand = DeviceRand(cfg.SEED, first_stream=(1 << 20) + (it << 12), device=dev)
        from od_wscl_amd.layers.misc import library_reference
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp), library_reference():     # MIOpen convolutions: the comparison
            losses, _ = model(images, targets, rois, rand=rand)
        opt.zero_grad(set_to_none=True)
        sum(losses.values()).float().backward()
        opt.step()

    for it in range(args.warmup):
        step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        step(args.warmup + it)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"comparison": "torch / hipBLASLt / MIOpen / torch.optim.SGD (%s)" % args.dtype,
                      "proposals_per_s": round(args.proposals * args.steps / dt, 1),
                      "ms_per_step": round(dt / args.steps * 1e3, 3)}))


if __name__ == "__main__":
    main()
