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
    # 3) the update the fused kernel performs with grad_scale = 1/world (torch restatement)
    p = torch.from_numpy(synthetic.fill_weights(3, 9, (n,), 0.1).copy())
    buf = torch.zeros(n)
    lr, wd, mu = 0.01, 1e-4, 0.9
    d = g * (1.0 / world) + wd * p
    buf = d.clone()
    p_new = p - lr * buf
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), boxes=rois[0].bbox.numpy(), labels=targets[0].get_field("labels").numpy(),
             img_sum=float(images.tensors.sum()), w=w, g=g.numpy(), local=local.numpy(), p_new=p_new.numpy())
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


def test_world_one_is_a_noop():
    from od_wscl_amd import engine
    g = torch.ones(10)
    engine.all_reduce_flat(g, 1)
    assert torch.equal(g, torch.ones(10))
