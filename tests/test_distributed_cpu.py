"""The N>1 path on CPU: two gloo processes (world_size 2).

What shards: images (one per rank, SOLVER.IMS_PER_BATCH // world in the reference,
data/build.py:150-155); what is exchanged: one sum-all-reduce of the flat gradient buffer, the
mean's 1/world folded into the SGD step.  The HIP kernels themselves need a GPU; here the
collective logic, the rank -> data mapping and the update arithmetic are checked."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from od_wscl_amd import engine, synthetic
    import bench
    # 1) rank -> data: each rank draws its own image / proposals / labels, identical weights
    images, targets, rois = bench.synthetic_batch(1234, rank, 96, 40, 21, torch.device("cpu"))
    w = synthetic.fill_weights(1, 1000, (64, 32), 0.01)
    # 2) the exchange: chunked in-place all-reduce of a flat buffer (3 chunks + ragged tail)
    n = 1000003
    g = torch.from_numpy(synthetic.fill_weights(7 + rank, 5, (n,), 1.0).copy())
    local = g.clone()
    engine.all_reduce_flat(g, world, chunk_elems=400000)
    # 2b) the same exchange piece by piece, as the pieces become final (engine.GradExchange: what the training step
    #     runs -- weight-gradient row blocks announced from backward, the rest swept up at the end), fp32 and bf16 wire
    gx = local.clone()
    ex = engine.GradExchange(gx, world, chunk_elems=150000)
    ex.begin()
    ex.ready(300000, 700000)          # e.g. fc6's gradient, block by block
    ex.ready(700000, 700100)
    ex.ready(10, 1234)
    left = ex.pending(0, n)
    ex.finish(0, 900000)              # the head ...
    ex.finish(900000, n)              # ... then the backbone and the biases
    g16 = local.clone()
    try:
        e16 = engine.GradExchange(g16, world, dtype="bf16", chunk_elems=400000)
        e16.begin()
        e16.ready(0, 500000)
        e16.finish(0, n)
        bf16_ok = True
    except RuntimeError:              # a gloo build without bf16 reductions: the wire format is exercised on the GPU box
        bf16_ok = False
    # 3) the update the fused kernel performs with grad_scale = 1/world (torch restatement)
    p = torch.from_numpy(synthetic.fill_weights(3, 9, (n,), 0.1).copy())
    buf = torch.zeros(n)
    lr, wd, mu = 0.01, 1e-4, 0.9
    d = g * (1.0 / world) + wd * p
    buf = d.clone()
    p_new = p - lr * buf
    # 4) the logging reduce of the reference's trainer (engine/trainer.py:14-36)
    red = engine.reduce_loss_dict({"loss_b": torch.tensor(float(rank + 1)), "loss_a": torch.tensor(10.0 * (rank + 1))}, world)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), red_a=float(red["loss_a"]), red_b=float(red["loss_b"]), boxes=rois[0].bbox.numpy(), labels=targets[0].get_field("labels").numpy(),
             img_sum=float(images.tensors.sum()), w=w, g=g.numpy(), local=local.numpy(), p_new=p_new.numpy(), gx=gx.numpy(),
             left=np.array(left), g16=g16.numpy(), bf16_ok=np.array(int(bf16_ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_and_sharding(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert not np.array_equal(r0["boxes"], r1["boxes"]) and r0["img_sum"] != r1["img_sum"]   # different images
    np.testing.assert_array_equal(r0["w"], r1["w"])                                         # same weights
    np.testing.assert_array_equal(r0["g"], r1["g"])                                         # all-reduced
    np.testing.assert_allclose(r0["g"], r0["local"] + r1["local"], rtol=1e-6, atol=1e-6)   # = sum over ranks
    np.testing.assert_array_equal(r0["p_new"], r1["p_new"])                                 # replicas stay in sync
    np.testing.assert_array_equal(r0["gx"], r0["g"])               # piecewise schedule == the monolithic exchange, bit for bit
    np.testing.assert_array_equal(r0["gx"], r1["gx"])
    assert r0["left"].tolist() == [[0, 10], [1234, 300000], [700100, 1000003]]
    if int(r0["bf16_ok"]):                                         # bf16 on the wire: both ranks agree, within bf16 of the sum
        np.testing.assert_array_equal(r0["g16"], r1["g16"])
        want = r0["local"] + r1["local"]
        assert np.abs(r0["g16"] - want).max() <= 2.0 ** -7 * np.abs(want).max()
    assert float(r0["red_a"]) == 15.0 and float(r0["red_b"]) == 1.5                          # rank 0: mean over the ranks


def test_world_one_is_a_noop():
    from od_wscl_amd import engine
    g = torch.ones(10)
    engine.all_reduce_flat(g, 1)
    assert torch.equal(g, torch.ones(10))
    ex = engine.GradExchange(g, 1, dtype="bf16")
    ex.begin(); ex.ready(0, 5); ex.finish(0, 10)
    assert torch.equal(g, torch.ones(10)) and ex.stage is None
    d = {"loss": torch.tensor(2.0)}
    assert engine.reduce_loss_dict(d, 1) is d


def test_lr_schedule_and_momentum_correction():
    """engine.lr_factor == WarmupMultiStepLR (solver/lr_scheduler.py:14-56); momentum_correction == update_momentum's
    rule (engine/trainer.py:38-51)."""
    from od_wscl_amd import engine
    from od_wscl_amd.config import make_defaults
    cfg = make_defaults()
    cfg.merge_from_list(["SOLVER.STEPS", (20, 30), "SOLVER.WARMUP_ITERS", 10, "SOLVER.BASE_LR", 0.01])
    assert abs(engine.lr_factor(cfg, 0) - 1.0 / 3) < 1e-12
    assert abs(engine.lr_factor(cfg, 5) - (1.0 / 3 * 0.5 + 0.5)) < 1e-12
    assert engine.lr_factor(cfg, 10) == 1.0 and engine.lr_factor(cfg, 19) == 1.0
    assert abs(engine.lr_factor(cfg, 20) - 0.1) < 1e-12 and abs(engine.lr_factor(cfg, 30) - 0.01) < 1e-12
    assert engine.momentum_correction(0.01, 0.0101) is None                    # < 10 % change
    assert abs(engine.momentum_correction(0.01, 0.001) - 0.1) < 1e-12          # milestone
    assert engine.momentum_correction(1e-8, 0.01) is None                      # cur_lr <= 1e-7
    if os.path.isdir("/root/reference/wetectron"):                            # the reference class itself, when present
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import refimport
        refimport.load_reference()
        from wetectron.solver.lr_scheduler import WarmupMultiStepLR
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=0.01)
        sch = WarmupMultiStepLR(opt, (20, 30), 0.1, warmup_factor=1.0 / 3, warmup_iters=10, warmup_method="linear")
        for it in range(1, 40):
            opt.step()
            sch.step()
            assert abs(opt.param_groups[0]["lr"] - 0.01 * engine.lr_factor(cfg, it)) < 1e-12, it


def test_checkpoint_layout_and_suffix_matching(tmp_path):
    """utils/checkpoint: the reference's `.pth` layout, the "module." prefix of DDP checkpoints and the
    longest-suffix matching of pretrained backbone keys (utils/model_serialization.py:11-80)."""
    from od_wscl_amd.utils import checkpoint as ck

    class Body(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(3, 4, 3)
            self.layer1 = torch.nn.Sequential(torch.nn.Conv2d(4, 4, 1))

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Sequential()
            self.backbone.add_module("body", Body())
            self.head = torch.nn.Linear(4, 2)

    a, b = Net(), Net()
    ck.save_checkpoint(a, str(tmp_path / "m.pth"), iteration=7)
    rest = ck.load_checkpoint(b, str(tmp_path / "m.pth"))
    assert rest["iteration"] == 7
    for (n1, p1), (n2, p2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2)
    # DDP prefix + a pretrained body with short keys: "conv1.weight" must go to body.conv1, "layer1.0.weight" to layer1
    c = Net()
    short = {"module.conv1.weight": torch.full((4, 3, 3, 3), 2.0), "module.layer1.0.weight": torch.full((4, 4, 1, 1), 3.0)}
    matched = ck.load_state_dict(c, short)
    assert matched == {"backbone.body.conv1.weight": "conv1.weight", "backbone.body.layer1.0.weight": "layer1.0.weight"}
    assert (c.backbone.body.conv1.weight == 2).all() and (c.backbone.body.layer1[0].weight == 3).all()
    torch.save({"w": 1}, str(tmp_path / "bare.pth"))            # a bare state-dict is wrapped as {"model": ...}
    assert ck.load_checkpoint(Net(), str(tmp_path / "bare.pth")) == {}


def test_optimizer_state_round_trips_in_the_reference_layout(tmp_path):
    """FlatSGD's momenta are saved as a torch.optim.SGD state_dict (one group per trainable parameter, the layout
    the reference's Checkpointer writes, utils/checkpoint.py:41-63 + solver/build.py:10-24): torch's own optimizer
    loads it, and a resumed FlatSGD continues with the same momenta, `first` cleared and the schedule position."""
    from od_wscl_amd import engine
    from od_wscl_amd.config import make_defaults
    from od_wscl_amd.utils import checkpoint as ck

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 4, 3)
            self.frozen = torch.nn.Conv2d(4, 4, 1)
            self.out = torch.nn.Conv2d(4, 2, 1)
            for p in self.frozen.parameters():
                p.requires_grad_(False)

    cfg = make_defaults()
    cfg.merge_from_list(["SOLVER.BASE_LR", 0.01, "SOLVER.STEPS", (5, 8), "SOLVER.WARMUP_ITERS", 2])
    torch.manual_seed(0)
    a = Net()
    oa = engine.FlatSGD(cfg, a, world=1)
    oa.flat_m.copy_(torch.randn_like(oa.flat_m))
    oa.first = False
    oa.lr_scale = engine.lr_factor(cfg, 6)
    path = str(tmp_path / "model_0000006.pth")
    ck.save_checkpoint(a, path, optimizer=oa, iteration=6)
    ck.tag_last_checkpoint(str(tmp_path), path)
    assert ck.last_checkpoint(str(tmp_path)) == path and ck.last_checkpoint(str(tmp_path / "nope")) is None
    saved = torch.load(path)
    assert set(saved) == {"model", "optimizer", "scheduler", "iteration"} and saved["scheduler"]["last_epoch"] == 6
    # torch.optim.SGD with the reference's grouping accepts it
    trainable = [(n, p) for n, p in Net().named_parameters() if p.requires_grad]
    ref_opt = torch.optim.SGD([{"params": [p], "lr": 0.01} for n, p in trainable], 0.01, momentum=0.9)
    ref_opt.load_state_dict(saved["optimizer"])
    assert len(ref_opt.state_dict()["state"]) == len(trainable)
    assert abs(saved["optimizer"]["param_groups"][1]["lr"] - 0.01 * 2 * 0.1) < 1e-12      # bias: lr x2, after one decay step
    # resume into a fresh model + optimizer
    b = Net()
    ob = engine.FlatSGD(cfg, b, world=1)
    assert ob.first and float(ob.flat_m.abs().sum()) == 0.0
    rest = ck.load_checkpoint(b, path)
    assert ck.restore_training_state(ob, b, rest) == 6
    assert not ob.first and ob.lr_scale == oa.lr_scale
    for n, (off, k) in oa.slices.items():
        o2, _ = ob.slices[n]
        assert torch.equal(oa.flat_m[off:off + k], ob.flat_m[o2:o2 + k]), n
        assert torch.equal(oa.flat_p[off:off + k], ob.flat_p[o2:o2 + k]), n
