"""Host-side logic added in round 3, checked without a GPU: the runs of layers the body's backward is cut into for the
data-parallel exchange, the graph walk that finds the leaves an early backward must feed by hand, the weight-gradient
batch's late registrations."""
import types

import torch

from od_wscl_amd import gemm
from od_wscl_amd.modeling.backbone.vgg16_hip import backward_segments
from od_wscl_amd.modeling.roi_heads.weak_head.loss_fused import _leaves_between


def _net(trainable, segments):
    return types.SimpleNamespace(layers=[types.SimpleNamespace(trainable=t) for t in trainable], bwd_segments=segments)


def test_backward_segments_cover_the_trainable_layers_in_backward_order():
    vgg = [False] * 4 + [True] * 9                      # conv1_x, conv2_x frozen (FREEZE_CONV_BODY_AT = 2)
    assert backward_segments(_net(vgg, 1)) == [(12, 4)]
    assert backward_segments(_net(vgg, 3)) == [(12, 10), (9, 7), (6, 4)]          # conv5 / conv4 / conv3
    assert backward_segments(_net(vgg, 4)) == [(12, 10), (9, 7), (6, 4)]          # ceil(9 / 4) = 3 layers per run
    assert backward_segments(_net(vgg, 99)) == [(i, i) for i in range(12, 3, -1)]
    two = backward_segments(_net([True] * 5, 2))
    assert two == [(4, 2), (1, 0)]
    for segs, n in ((backward_segments(_net(vgg, k)), 9) for k in range(1, 12)):
        covered = [li for hi, lo in segs for li in range(hi, lo - 1, -1)]
        assert covered == list(range(12, 3, -1)) and len(covered) == n            # every layer once, descending


def test_leaves_between_stops_at_the_cut_and_finds_cat_ed_parameters():
    x = torch.randn(4, 3, requires_grad=True)
    pooled = x * 2.0                                     # stands for the pooling node's output (the cut)
    w1, w2, b = (torch.randn(3, 3, requires_grad=True) for _ in range(3))
    frozen = torch.randn(3, 3)
    w_cat = torch.cat([w1, w2], dim=0)                   # the eight predictor heads behind a torch.cat
    y = (pooled @ w_cat.t()).sum() + (pooled @ b).sum() + (pooled @ frozen).sum()
    leaves = _leaves_between(y.grad_fn, pooled.grad_fn)
    assert {id(t) for t in leaves} == {id(w1), id(w2), id(b)}                     # not x (behind the cut), not `frozen`
    grads = torch.autograd.grad(y, [pooled] + leaves, allow_unused=True)
    assert grads[0].shape == pooled.shape and all(g is not None for g in grads[1:])


def test_wgrad_batch_takes_registrations_after_its_first_block():
    from od_wscl_amd import precision
    old, mode = gemm.WgradBatch.reserve, precision.get_precision()
    try:
        precision.set_precision("bf16x2f")               # single-plane backward: one column block per evaluation
        gemm.WgradBatch.reserve = 128
        b = gemm.WgradBatch()
        s0, s1 = b.register(200), b.register(70)         # rounded to 256 and 128 columns
        assert (b.offset(s0), b.offset(s1)) == (0, 256)
        dzt, xt = b.buffers(8, 16, torch.device("cpu"))
        assert dzt.shape == (8, 384 + 128) and xt.shape == (16, 384 + 128) and b.kpad == 384 and b.done == [False, False]
        dzt[:, :384] = 1.0
        xt[:, :384] = 2.0
        s2 = b.register(100)                             # fits the reserve: same buffers, K grows
        assert b.offset(s2) == 384 and b.kpad == 512 and b.buffers(8, 16, torch.device("cpu"))[0] is dzt and b.done == [False] * 3
        s3 = b.register(64)                              # beyond it: new buffers, the filled columns copied
        d2, x2 = b.buffers(8, 16, torch.device("cpu"))
        assert b.offset(s3) == 512 and b.kpad == 576 and d2 is not dzt and d2.shape[1] >= 576
        assert torch.equal(d2[:, :384], torch.ones(8, 384, dtype=torch.bfloat16)) and torch.equal(x2[:, :384], torch.full((16, 384), 2.0, dtype=torch.bfloat16))
        b.reset()
        assert b.rows == [] and b.dzt is None and b.done == []
    finally:
        gemm.WgradBatch.reserve = old
        precision.set_precision(mode)
