"""The N>1 training step on the GPU box: two processes sharing the one MI355X, gloo as the transport (RCCL refuses
two ranks on one device; the collective CALLS, their stream ordering and the 1/world scaling are identical).
Each rank steps the full hot path on its own image; after every step the replicas' parameters must be bit-identical,
and the early (side-stream) all-reduce of the head must give the same result as the plain end-of-backward one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, overlap, dtype="bf16"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      ODW_NO_TIMER="1", ODW_NO_OVERLAP="0" if overlap else "1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from od_wscl_amd import engine
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = bench.build_cfg(21)
    step, _ = engine.build_training_step(cfg, dev, dtype=dtype, world=world, seed=cfg.SEED, backend="hip")
    images, targets, rois = bench.synthetic_batch(cfg.SEED, rank, 224, 120, 21, dev)
    fc6 = step.model.roi_heads.feature_extractor.fc6.weight
    # with the overlap on, the large weight gradients are handed to the exchange from backward, fc6's in row blocks
    assert (getattr(fc6, "_odw_grad_ready", None) is not None) == overlap and (not overlap or 0 < fc6._odw_slice_rows < fc6.shape[0])
    # ... and the body's backward runs in three runs of layers, each announcing its weight gradients to the exchange
    body = step.model.hip_body()
    assert body.bwd_segments == (3 if overlap else 1) and (body.on_segment_done is not None) == overlap
    losses = []
    for it in range(3):
        l, _ = step(images, targets, rois, DeviceRand(cfg.SEED + rank, first_stream=(1 << 20) + (it << 12), device=dev))
        losses.append(float(sum(l.values())))
    torch.cuda.synchronize()
    opt = step.optimizer
    np.savez(os.path.join(out_dir, "r%d_%d.npz" % (rank, int(overlap))), p=opt.flat_p.cpu().numpy(),
             m=opt.flat_m.cpu().numpy(), losses=np.array(losses), early=np.array(int(opt.early_done)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["bf16x2f", "bf16"])
def test_two_rank_step_keeps_replicas_identical(tmp_path, dtype):
    """overlap True: the weight gradients of fc6 / fc7 / Sim_Net are all-reduced as their GEMMs retire (fc6 in row
    blocks, the gradient GEMM itself cut to match), the rest of the head right after the pooling backward, the backbone
    after backward; overlap False: ONE exchange of the whole buffer after backward.  Same parameters either way."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = {}
    for overlap in (True, False):
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), overlap, dtype), nprocs=2, join=True)
        res[overlap] = [np.load(tmp_path / ("r%d_%d.npz" % (r, int(overlap)))) for r in range(2)]
    for overlap, (a, b) in res.items():
        assert np.isfinite(a["losses"]).all() and np.isfinite(b["losses"]).all()
        assert not np.array_equal(a["losses"], b["losses"])                    # different images per rank
        np.testing.assert_array_equal(a["p"], b["p"])                          # replicas in lock step
        np.testing.assert_array_equal(a["m"], b["m"])
        assert int(a["early"]) == int(overlap)
    # early exchange of the head on the side stream == one exchange after backward, up to the run-to-run noise of
    # the atomic accumulations in the ROI pooling backward (~1e-7 absolute after three steps; a flipped near-tie selection moves a parameter by at most lr x |grad|)
    np.testing.assert_allclose(res[True][0]["p"], res[False][0]["p"], rtol=0, atol=1e-4)


def _rccl_worker(rank, port, out_dir):
    """One rank, backend "nccl" (= RCCL): the collectives degenerate to copies, but the process group, the chunked
    async all-reduce on the side stream and the bench's barrier / MAX reduction run through RCCL's real entry points."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", ODW_NO_TIMER="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from od_wscl_amd import engine
    from od_wscl_amd.utils.device_rand import DeviceRand
    flat = torch.arange(3_000_000, dtype=torch.float32, device=dev)
    want = flat.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        engine.all_reduce_flat(flat, 2, chunk_elems=1 << 20)         # world=2 forces the collective path; the group has 1 rank
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(flat, want)
    vals = []
    for wire in ("fp32", "bf16"):      # bf16 on the wire: RCCL's bf16 sum through the staging buffer, fp32 back
        cfg = bench.build_cfg(21)
        cfg.merge_from_list(["ODW.GRAD_EXCHANGE", wire])
        step, _ = engine.build_training_step(cfg, dev, dtype="bf16x2f", world=2, seed=cfg.SEED, backend="hip")   # grad_scale 1/2
        assert step.optimizer.exchange.dtype == wire and (step.optimizer.exchange.stage is not None) == (wire == "bf16")
        images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 224, 120, 21, dev)
        for it in range(2):
            l, _ = step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20) + (it << 12), device=dev))
            vals.append(float(sum(l.values())))
        del step
    dist.barrier()
    t = torch.tensor([1.5], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rccl.npz"), losses=np.array(vals), t=t.cpu().numpy())
    dist.destroy_process_group()


def test_rccl_backend_single_rank_step(tmp_path):
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    assert np.isfinite(r["losses"]).all() and float(r["t"][0]) == 1.5


def test_bench_launches_two_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` end to end (tools/train_net.py:286-294 of the reference: one process per GPU): bench.py's
    own launcher (launch_ranks -> python -m torch.distributed.run, the path the driver's multi-GPU run takes when it does
    not start the ranks itself), the env:// rendezvous, the barrier / MAX-over-ranks timing, the early gradient exchange
    and the JSON line of rank 0.  One MI355X here, so the two ranks share it (--oversubscribe) and gloo carries the
    exchange (RCCL refuses two ranks on one device); the line says so and is not a scaling number."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--backend", "gloo",
           "--oversubscribe", "--proposals", "300", "--size", "224", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world_size"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["global_batch"] == 2 and out["scaling"] == "weak" and out["steps"] == 3 and out["warmup"] == 2
    assert out["value"] > 0 and abs(out["value"] - 2 * 300 * 3 / (out["ms_per_step"] * 3e-3)) <= 1e-2 * out["value"]
    assert abs(out["per_gpu"] * 2 - out["value"]) <= 1.0
    c = out["collective"]
    assert c["backend"] == "gloo" and c["ranks"] == 2 and c["wire_dtype"] == "fp32" and c["exposed_ms_per_step"] >= 0.0
    assert c["devices_shared"] and "oversubscribed" in out
    assert "secondary" not in out and "cpu_baseline" not in out           # N = 1 only
    # ... and the same launcher refuses to report an N-GPU number from fewer devices when not asked to share them
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1), "--steps", "1"],
                        env=env, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "refusing" in r2.stderr
