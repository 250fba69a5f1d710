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


# ---- round 4 ---------------------------------------------------------------------------------------------------------
def test_graph_cache_policy_is_bounded_and_rate_limited(monkeypatch):
    """VGGBackboneHip._graph_for without a GPU (the capture itself is stubbed): a shape runs eagerly the first time, is
    captured when it comes back, at most `graph_cache_size` shapes stay captured (least recently used evicted), and a
    workload whose shapes cycle faster than the cache holds stops capturing (2 + replays / 8) instead of re-capturing
    every step (configs/voc/*.yaml:33-34: six training scales x free aspect ratios)."""
    from od_wscl_amd.modeling.backbone import vgg16_hip as V

    class FakeGraphed(object):
        def __init__(self, net, fn, images):
            self.bytes = 0

    monkeypatch.setattr(V, "_GraphedBody", FakeGraphed)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda dev=None: 0)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: types.SimpleNamespace(synchronize=lambda: None))
    net = object.__new__(V.VGGBackboneHip)
    torch.nn.Module.__init__(net)
    import collections
    net._graphs, net._sightings = collections.OrderedDict(), {}
    net.graph_stats = {"captures": 0, "replays": 0, "eager": 0, "evictions": 0, "bytes": 0}
    net.graph_cache_size = 2

    def fn():
        pass

    def see(h, w):
        return net._graph_for(fn, torch.empty(1, 3, h, w))

    assert see(608, 608) is None                                  # first sighting: eager
    g = see(608, 608)
    assert g is not None and net.graph_stats["captures"] == 1     # comes back: captured
    assert all(see(608, 608) is g for _ in range(20))             # the bench's fixed shape: replays from then on
    assert see(480, 640) is None and see(480, 640) is not None    # a second shape
    assert len(net._graphs) == 2
    for _ in range(3):
        see(608, 608)
    assert see(800, 1216) is None
    assert see(800, 1216) is not None                             # third shape: the least recently used one (480 x 640) goes
    keys = [k[1][2:] for k in net._graphs]
    assert keys == [(608, 608), (800, 1216)] and net.graph_stats["evictions"] == 1
    # 40 distinct shapes, each seen twice in a row, with no replays in between: captures stop at the rate limit
    before = net.graph_stats["captures"]
    for i in range(40):
        see(480 + 32 * i, 640)
        see(480 + 32 * i, 640)
    st = net.graph_stats
    assert len(net._graphs) <= 2
    assert st["captures"] - before <= 2 + st["replays"] // 8
    assert st["eager"] >= 70


def test_grad_exchange_refuses_a_range_announced_twice(monkeypatch):
    """A flat range handed to the all-reduce twice in one step means its producer wrote into it after the first
    announcement (a weight-gradient batch flushed twice): the first collective summed a partial gradient."""
    import pytest
    from od_wscl_amd import engine
    ex = engine.GradExchange(torch.zeros(1024), world=2)
    monkeypatch.setattr(ex, "_issue", lambda lo, hi: None)
    ex.begin()
    ex.ready(0, 256)
    ex.ready(512, 1024)
    assert ex.pending(0, 1024) == [(256, 512)]
    with pytest.raises(RuntimeError, match="overlaps"):
        ex.ready(128, 300)
    ex.begin()                                                   # a new step starts clean
    ex.ready(128, 300)


def test_schedule_position_is_the_scheduler_step_count_not_the_iteration():
    """SOLVER.ITER_SIZE = 4: the reference steps its scheduler once per group (engine/trainer.py:86-91) and saves
    scheduler.last_epoch; the checkpointed `iteration` counts micro-iterations.  Resuming must take the learning-rate
    factor from the group count -- from the micro count it would cross a STEPS milestone 4x early."""
    from od_wscl_amd import engine
    from od_wscl_amd.config import make_defaults
    from od_wscl_amd.utils import checkpoint as ck
    cfg = make_defaults()
    cfg.merge_from_list(["SOLVER.ITER_SIZE", 4, "SOLVER.STEPS", (100, 200), "SOLVER.WARMUP_ITERS", 10, "SOLVER.GAMMA", 0.1])
    calls = []

    class Opt(engine.FlatSGD):
        def __init__(self):
            self.cfg, self.sched_steps, self.lr_scale, self.grads_clean = cfg, 0, 1.0, False

        def sync_from_params(self, model=None):
            calls.append("sync")

        def load_state_dict(self, model, sd):
            calls.append("momenta")

    opt = Opt()
    opt.sched_steps = 60                                          # 240 micro-iterations = 60 groups
    state = opt.scheduler_state(240)
    assert state["last_epoch"] == 60
    fresh = Opt()
    assert fresh.scheduler_state(240)["last_epoch"] == 60         # a caller that never numbered its steps: ceil(240 / 4)
    assert fresh.scheduler_state(241)["last_epoch"] == 61
    it = ck.restore_training_state(fresh, None, {"iteration": 240, "optimizer": {}, "scheduler": state})
    assert it == 240 and fresh.sched_steps == 60 and fresh.grads_clean and calls == ["momenta", "sync"]
    assert fresh.lr_scale == engine.lr_factor(cfg, 60) == 1.0     # 60 < 100: before the first milestone
    assert engine.lr_factor(cfg, 240) == 0.1 * 0.1                # what the micro count would have given
    legacy = Opt()                                                # a checkpoint without a scheduler entry
    ck.restore_training_state(legacy, None, {"iteration": 402})
    assert legacy.sched_steps == 101 and abs(legacy.lr_scale - 0.1) < 1e-12


def test_inject_grad_routes_the_early_gradient_through_every_backward_entry():
    """loss_fused._InjectGrad: once the dense losses' backward has run early, loss_sim carries their gradient w.r.t. the
    stacked operand, so total.backward() and sum(losses.values()).backward() deliver it like finish_backward()."""
    from od_wscl_amd.modeling.roi_heads.weak_head.loss_fused import _InjectGrad
    for entry in ("total", "sum", "loss_sim"):
        body = torch.randn(5, 3, requires_grad=True)
        x = body * 1.0                                            # the stacked operand (output of the pooling node)
        e = torch.randn(4, requires_grad=True)
        loss_sim = (e ** 2).sum()
        dx = torch.full((5, 3), 0.25)
        dense = torch.tensor([1.0, 2.0])                          # detached dense losses
        ls = _InjectGrad.apply(loss_sim, x, dx)
        losses = {"loss_img": dense[0], "loss_ref": dense[1], "loss_sim": ls}
        if entry == "total":
            (dense.sum() + ls).backward()
        elif entry == "sum":
            sum(losses.values()).backward()
        else:
            ls.backward()
        assert torch.equal(body.grad, dx) and torch.allclose(e.grad, 2 * e.detach())
        assert float(ls) == float(loss_sim)


def test_conv2d_has_no_library_forward_outside_the_reference_context():
    """layers.Conv2d owns parameters; the bodies' convolutions run in GeneralizedRCNN.hip_body().  A direct call must not
    reach torch's convolution (MIOpen on the GPU) silently: it raises, unless a test opts in for a reference."""
    import pytest
    from od_wscl_amd.layers import Conv2d
    from od_wscl_amd.layers.misc import library_reference
    conv = Conv2d(3, 4, kernel_size=3, padding=1)
    x = torch.randn(1, 3, 8, 8)
    with pytest.raises(RuntimeError, match="no standalone forward"):
        conv(x)
    with library_reference():
        y = conv(x)
    assert y.shape == (1, 4, 8, 8)
    with pytest.raises(RuntimeError):
        conv(x)


def test_cell_major_layout_is_kept_only_for_shapes_the_kernel_walks(monkeypatch):
    """layers.Linear.cm_layout -> gemm.Shadow.cm (the cell-major forward planes of the first head Linear): only
    for channel counts that are multiples of the 64-wide K tile and at most 64 cells (the keep mask is a 64-bit set); any other
    pooler resolution keeps the channel-major planes, and ODW_NO_PAIR=1 switches the form off altogether."""
    from od_wscl_amd.layers.linear import Linear
    ok = Linear(512 * 49, 8)
    ok.cm_layout = (512, 49)
    assert ok._get_shadow().cm == (512, 49) and ok.can_pair(512, 49) and not ok.can_pair(512, 36)
    big = Linear(64 * 196, 8)
    big.cm_layout = (64, 196)                      # a 14 x 14 pooler: more cells than the mask holds
    assert big._get_shadow().cm is None and not big.can_pair(64, 196)
    odd = Linear(48 * 49, 8)
    odd.cm_layout = (48, 49)
    assert odd._get_shadow().cm is None
    monkeypatch.setenv("ODW_NO_PAIR", "1")
    off = Linear(512 * 49, 8)
    off.cm_layout = (512, 49)
    assert off._get_shadow().cm is None and not off.can_pair(512, 49)
