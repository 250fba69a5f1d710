"""A training step that dies mid-backward leaves no state behind (engine.build_training_step's recovery path):
the next step runs, and gives what a step on an untouched engine gives.  -m gpu."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _engine():
    import bench
    from od_wscl_amd import engine
    dev = torch.device("cuda", 0)
    cfg = bench.build_cfg(21)
    step, _ = engine.build_training_step(cfg, dev, dtype="bf16x2f", world=1, seed=cfg.SEED)
    batch = bench.synthetic_batch(cfg.SEED, 1, 224, 160, 21, dev)       # (image 1: two labels)
    return cfg, dev, step, batch


def _losses(out):
    return {k: float(v) for k, v in out[0].items()}


def test_a_step_that_fails_in_backward_does_not_poison_the_next():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd.utils.device_rand import DeviceRand
    cfg, dev, step, (images, targets, rois) = _engine()
    rand = lambda it: DeviceRand(cfg.SEED, first_stream=(1 << 20) + (it << 12), device=dev)
    ref0 = _losses(step(images, targets, rois, rand(0), iteration=1))
    torch.cuda.synchronize()
    heads = step.model.roi_heads
    fe = heads.feature_extractor
    opt = step.optimizer
    # ---- the failure: the callback the pooling node's gradient hook runs (the head's early update) raises -- by then the
    # contrastive branch, the dense losses' early backward and the pooling node's backward have all run
    good = heads.head_grads_ready

    def boom():
        raise RuntimeError("injected failure in backward")
    heads.head_grads_ready = boom
    sched_before = opt.sched_steps
    with pytest.raises(RuntimeError, match="injected failure"):
        step(images, targets, rois, rand(1), iteration=2)
    heads.head_grads_ready = good
    assert opt.sched_steps == sched_before, "the failed iteration's scheduler step was not rolled back"
    assert opt.grads_clean and not getattr(opt, "hold", False)
    assert fe._grad_holder is None or not fe._grad_holder.pending
    for sh in opt.shadows:
        b = getattr(sh, "batch", None)
        assert b is None or (not b.rows and b.dzt is None)
    # ---- the retried iteration runs and equals the same iteration on an engine that never failed
    got = _losses(step(images, targets, rois, rand(1), iteration=2))
    torch.cuda.synchronize()
    del step
    cfg2, dev2, step2, _ = _engine()
    r0 = _losses(step2(images, targets, rois, rand(0), iteration=1))
    want = _losses(step2(images, targets, rois, rand(1), iteration=2))
    assert r0 == ref0                                    # (deterministic kernels: the first step is the first step)
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-6 * max(abs(want[k]), 1e-6), (k, got[k], want[k])


def test_a_step_that_overflows_the_sampled_row_capacity_raises_soon_after():
    """ODW.MAX_SAMPLED_ROWS bounds the IoU-sampled rows of a step on the device-resident path; a step that samples more sets a
    sticky device flag (csrc/loss_lists.hip never writes past a buffer) and a later step's non-blocking look at the flags
    raises -- loudly, not silently on truncated lists."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd.utils.device_rand import DeviceRand
    cfg, dev, step, (images, targets, rois) = _engine()
    step.model.roi_heads.loss_evaluator.max_sampled_rows = 8          # (every group of this batch samples more)
    rand = lambda it: DeviceRand(cfg.SEED, first_stream=(1 << 20) + (it << 12), device=dev)
    with pytest.raises(RuntimeError, match="did not fit its buffers"):
        for it in range(6):
            step(images, targets, rois, rand(it), iteration=it + 1)
            torch.cuda.synchronize()
