"""One training step of the hot path (the caller of GeneralizedRCNN.forward in the reference:
engine/trainer.py:79-120 -- forward, sum of losses, backward, DDP all-reduce, SGD step)."""
import torch
import torch.distributed as dist

from . import synthetic
from .modeling.detector import build_detection_model
from .utils.kernel_timer import kernel_timer  # noqa: F401  (re-exported for bench.py)


def load_formula_weights(model, seed, overrides=None):
    shapes = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    sd = synthetic.init_state_dict(shapes, seed, overrides=overrides)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(sd[n]))


def make_optimizer(cfg, model):
    """solver/build.py:10-24: one group per parameter; biases get lr x BIAS_LR_FACTOR and
    WEIGHT_DECAY_BIAS."""
    groups = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
        if "bias" in key:
            lr, wd = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR, cfg.SOLVER.WEIGHT_DECAY_BIAS
        groups.append({"params": [value], "lr": lr, "weight_decay": wd})
    return torch.optim.SGD(groups, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)


def build_training_step(cfg, device, dtype="bf16", world=1, seed=1234):
    model = build_detection_model(cfg).to(device)
    load_formula_weights(model, 1)
    model.train()
    fe = model.roi_heads.feature_extractor
    fe.classifier[1].tag = "fc6"
    fe.classifier[4].tag = "fc7"
    kernel_timer.enabled = True
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index], broadcast_buffers=False,
                                                        bucket_cap_mb=128, gradient_as_bucket_view=True)
    opt = make_optimizer(cfg, model)
    use_autocast = dtype == "bf16"
    model.roi_heads.loss_evaluator.amp = use_autocast

    import os
    debug = os.environ.get("ODW_DEBUG_SYNC", "").split(",")
    kernel_timer.enabled = os.environ.get("ODW_NO_TIMER") != "1"

    def mark(tag):
        if tag in debug or "all" in debug:
            torch.cuda.synchronize()
            print("[odw] ok", tag, flush=True)

    def step(images, targets, rois, rand):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_autocast):
            losses, accs = net(images, targets, rois, rand=rand)
        mark("forward")
        loss = sum(losses.values())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        mark("backward")
        opt.step()
        mark("optimizer")
        if "loss" in debug:
            print("[odw] losses", {k: round(float(v), 5) for k, v in losses.items()}, flush=True)
        return losses, accs

    info = {"gemm_backend": "torch/hipBLASLt (%s)" % dtype, "conv_backend": "torch/MIOpen (%s)" % dtype}
    return step, info
