"""End-to-end parity on the GPU: the product model (od_wscl_amd.modeling, gfx950 kernels: HIP body, MFMA Linear
layers, fused loss) against the golden vectors the REFERENCE produced on the same formula-generated inputs.
`pytest -m gpu`.  The parity bars are asserted in the fp32-grade precision mode "bf16x3" (tests/conftest.py sets it for
every test): the same kernels, the same orchestration as the throughput mode, operands carried as three bf16 planes.

Bars (BASELINE.json north_star): ROI / NMS / pseudo-label index selection bit-exact, fp32 losses
within 1e-3 relative."""
import numpy as np
import pytest
import torch

from conftest import e2e_arch, e2e_classes, e2e_inputs, load_e2e, weights_for

pytestmark = pytest.mark.gpu

E2E = ["e2e_voc_2img", "e2e_voc_1img", "e2e_align_1img", "e2e_r50_2img"]
E2E_ALL = E2E + ["e2e_coco_2img"]          # the COCO14 shape: 81 classes, predictor N = 1377
LOSS_RTOL = 1e-3


ARCH_OPTS = {
    "vgg16": ["MODEL.BACKBONE.CONV_BODY", "VGG16-OICR", "MODEL.ROI_BOX_HEAD.POOLER_SCALES", (0.125,),
              "MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR", "VGG16.roi_head"],
    # configs/voc/voc07_r50_c5_contra_db_b8_lr0.02_ss.yaml
    "r50": ["MODEL.BACKBONE.CONV_BODY", "R-50-C5", "MODEL.ROI_BOX_HEAD.POOLER_SCALES", (0.0625,),
            "MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR", "ResNet50Conv5ROIFeatureExtractor"],
}


def build_model(pooler, weights_np, loss_impl="fused", arch="vgg16", classes=21):
    from od_wscl_amd.config import make_defaults
    from od_wscl_amd.modeling.detector import build_detection_model
    cfg = make_defaults()
    cfg.merge_from_list(["MODEL.WSOD_ON", True, "MODEL.FASTER_RCNN", False,
                         "MODEL.ROI_BOX_HEAD.NUM_CLASSES", classes, "MODEL.ROI_BOX_HEAD.POOLER_METHOD", pooler,
                         "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", 7,
                         "MODEL.ROI_WEAK_HEAD.REGRESS_ON", True, "DB.METHOD", "dropblock", "SOLVER.CONTRA", True,
                         "nms", 0.1, "lmda", 0.03, "temp", 0.2, "ODW.LOSS_IMPL", loss_impl] + ARCH_OPTS[arch])
    model = build_detection_model(cfg).cuda()
    with torch.no_grad():
        for n, p in list(model.named_parameters()) + list(model.named_buffers()):
            p.copy_(torch.from_numpy(weights_np[n]))
    model.train()
    return model


@pytest.mark.parametrize("loss_impl", ["fused", "loops"])
@pytest.mark.parametrize("name", E2E_ALL)
def test_model_matches_reference_golden(name, loss_impl):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd.structures import BoxList, to_image_list
    from od_wscl_amd.utils.device_rand import DeviceRand
    g = load_e2e(name)
    seed, batch, boxes, labels, cfg = e2e_inputs(g)
    model = build_model(cfg["pooler"], weights_for(e2e_arch(g), e2e_classes(g)), loss_impl, e2e_arch(g), e2e_classes(g))
    specs = g["spec_images"]
    rois, targets = [], []
    for k, (h, w, p) in enumerate(specs):
        rois.append(BoxList(boxes[k].cuda(), (int(w), int(h)), "xyxy"))
        t = BoxList(torch.zeros((len(labels[k]), 4)).cuda(), (int(w), int(h)), "xyxy")
        t.add_field("labels", labels[k].cuda())
        targets.append(t)
    images = to_image_list(batch.cuda())
    rand = DeviceRand(seed)
    trace = {}
    model.roi_heads.loss_evaluator.trace = trace
    losses, accs = model(images, targets, rois, iteration={"iter": 1}, rand=rand)
    assert rand.s.next == int(g["streams_used"])
    for k, v in losses.items():
        ref = float(g["loss/" + k])
        assert abs(float(v.detach()) - ref) <= LOSS_RTOL * max(abs(ref), 1e-6), (k, float(v.detach()), ref)
    for k, v in accs.items():
        assert abs(float(v) - float(g["acc/" + k])) < 1e-6, k
    for k in g.files:
        if k.startswith(("pseudo_", "pgt_instance_")):
            np.testing.assert_array_equal(trace[k].cpu().numpy(), g[k], err_msg=k)   # bit-exact selection
        if k.startswith("weights_"):
            np.testing.assert_allclose(trace[k].cpu().numpy(), g[k], rtol=1e-3, atol=1e-7)
    assert trace["supcon_n"] == int(g["supcon_n"])
    if loss_impl == "fused":
        assert trace.get("dense_loss_kernel"), "the fused dense-loss kernel was not taken"
    np.testing.assert_allclose(trace["supcon_weights"].cpu().numpy(), g["supcon_weights"], rtol=1e-3, atol=1e-9)
    sum(losses.values()).backward()
    for n, p in model.named_parameters():
        key = "gradnorm/" + n
        if key in g.files:
            ref = float(g[key])
            got = p.grad.double().norm().item()
            # analytically-zero gradients (e.g. det_score.bias: a column softmax ignores its bias) are pure
            # rounding noise ~1e-8 in both implementations -> absolute floor
            assert abs(got - ref) <= 2e-3 * ref + 1e-6, (n, got, ref)
        else:
            assert p.grad is None


def test_device_rng_matches_host():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd.utils import rng
    from od_wscl_amd.utils.device_rand import DeviceRand
    r = DeviceRand(99, first_stream=5)
    u = r.uniform((1000, 7, 7)).cpu().numpy().ravel()
    np.testing.assert_array_equal(u, rng.uniform(99, 5, 49000))          # bit-identical stream
    z = r.normal((333, 3)).cpu().numpy().ravel()
    np.testing.assert_allclose(z, rng.normal(99, 6, 999), rtol=0, atol=5e-6)
    x = torch.randn(257, 33, device="cuda", requires_grad=True)
    y = r.dropout(x, 0.5)
    keep = rng.uniform(99, 7, 257 * 33).reshape(257, 33) >= 0.5
    np.testing.assert_array_equal(y.detach().cpu().numpy(), x.detach().cpu().numpy() * keep * 2.0)
    y.sum().backward()
    np.testing.assert_array_equal(x.grad.cpu().numpy(), keep * 2.0)       # backward re-derives the mask
    x2 = torch.randn(64, 10, device="cuda", requires_grad=True)
    n = r.noise_mul(x2)
    zn = rng.normal(99, 8, 640).reshape(64, 10)
    np.testing.assert_allclose(n.detach().cpu().numpy(), x2.detach().cpu().numpy() * (1 + zn), rtol=1e-5, atol=1e-6)


def test_od_assign_kernel(ops_golden):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import _C
    g = ops_golden
    a = torch.from_numpy(g["iou_a"]).cuda()
    score = torch.from_numpy(g["od_score"]).cuda()
    prob = score[:, 1:].clone()
    gt_b, gt_c, gt_s = [], [], []
    for c, key in ((3, "od_pgt3"), (10, "od_pgt10")):
        col = prob[:, c]
        top = torch.argmax(col)
        picked = torch.from_numpy(g[key]).cuda()
        gt_b.append(a[picked]); gt_c.append(torch.full((picked.numel(),), c + 1, device="cuda")); gt_s.append(col[picked].clone())
        prob[top].fill_(0)
    pl, lw, rt = _C.od_assign(a, torch.cat(gt_b), torch.cat(gt_c), torch.cat(gt_s))
    np.testing.assert_array_equal(pl.cpu().numpy(), g["od_pseudo"])
    np.testing.assert_array_equal(lw.cpu().numpy(), g["od_weights"])
    np.testing.assert_allclose(rt.cpu().numpy(), g["od_targets"], rtol=1e-6, atol=1e-6)


def _run_golden(name, mode):
    """(losses, selection trace, model after backward, golden) of one e2e golden in one precision mode."""
    from od_wscl_amd import precision
    from od_wscl_amd.structures import BoxList, to_image_list
    from od_wscl_amd.utils.device_rand import DeviceRand
    g = load_e2e(name)
    seed, batch, boxes, labels, cfg = e2e_inputs(g)
    precision.set_precision(mode)
    model = build_model(cfg["pooler"], weights_for(e2e_arch(g), e2e_classes(g)), "fused", e2e_arch(g), e2e_classes(g))
    rois, targets = [], []
    for k, (h, w, p) in enumerate(g["spec_images"]):
        rois.append(BoxList(boxes[k].cuda(), (int(w), int(h)), "xyxy"))
        t = BoxList(torch.zeros((len(labels[k]), 4)).cuda(), (int(w), int(h)), "xyxy")
        t.add_field("labels", labels[k].cuda())
        targets.append(t)
    trace = {}
    model.roi_heads.loss_evaluator.trace = trace
    losses, accs = model(to_image_list(batch.cuda()), targets, rois, rand=DeviceRand(seed))
    sum(losses.values()).backward()
    return losses, trace, model, g


@pytest.mark.parametrize("name", E2E_ALL)
def test_bf16x2_mode_meets_the_loss_and_selection_bars(name):
    """Two bf16 planes per operand, three plane products (half the MFMA work of bf16x3): on all four goldens the losses
    stay within the 1e-3 bar and every selected index set is the reference's (observed: losses <= 9e-5, profiles/r02/
    precision_deviation.json); gradient norms within 5e-3 (observed <= 2.8e-3 -- the 2e-3 bar needs bf16x3)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    losses, trace, model, g = _run_golden(name, "bf16x2")
    for k, v in losses.items():
        ref = float(g["loss/" + k])
        assert abs(float(v.detach()) - ref) <= LOSS_RTOL * max(abs(ref), 1e-6), (k, float(v.detach()), ref)
    for k in g.files:
        if k.startswith(("pseudo_", "pgt_instance_")):
            np.testing.assert_array_equal(trace[k].cpu().numpy(), g[k], err_msg=k)
    for n, p in model.named_parameters():
        key = "gradnorm/" + n
        if key in g.files:
            ref = float(g[key])
            assert abs(p.grad.double().norm().item() - ref) <= 5e-3 * ref + 1e-6, n


MIXED_GRAD_TOL = 1e-2          # observed: <= 3.8e-3 on the five goldens


@pytest.mark.parametrize("name", E2E_ALL)
def test_bf16x2f_mode_meets_the_loss_and_selection_bars(name):
    """The bench's default mode since round 3: forward products on two bf16 planes (as "bf16x2"), backward products on
    one.  BASELINE.json's bar -- losses within 1e-3, every selected index set the reference's -- holds on all five
    goldens; the gradients (not part of that bar) carry the bf16 rounding of the backward operands."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    losses, trace, model, g = _run_golden(name, "bf16x2f")
    for k, v in losses.items():
        ref = float(g["loss/" + k])
        assert abs(float(v.detach()) - ref) <= LOSS_RTOL * max(abs(ref), 1e-6), (k, float(v.detach()), ref)
    for k in g.files:
        if k.startswith(("pseudo_", "pgt_instance_")):
            np.testing.assert_array_equal(trace[k].cpu().numpy(), g[k], err_msg=k)
    worst = 0.0
    for n, p in model.named_parameters():
        key = "gradnorm/" + n
        if key in g.files and float(g[key]) > 1e-5:
            ref = float(g[key])
            dev = abs(p.grad.double().norm().item() - ref) / ref
            worst = max(worst, dev)
            assert dev <= MIXED_GRAD_TOL, (n, dev)
    print("MIXEDREPORT", name, "worst gradient-norm deviation", worst,
          {k: (float(v.detach()), float(g["loss/" + k])) for k, v in losses.items()})


@pytest.mark.parametrize("name", ["e2e_voc_2img", "e2e_voc_1img", "e2e_coco_2img"])
@pytest.mark.parametrize("pair", [True, False])
def test_both_fc6_forward_forms_meet_the_bars(name, pair, monkeypatch):
    """The shared clean + DropBlock fc6 forward (gemm.pair_linear over cell-major planes: the default in "bf16x2f", here
    asserted to have run) and the stacked pass of rounds 1-3 (ODW_NO_PAIR=1) on the VGG / ROIPool goldens: losses within
    1e-3, every selected index set the reference's, gradient norms within the mode's tolerance."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import gemm
    if not pair:
        monkeypatch.setenv("ODW_NO_PAIR", "1")
    calls = []
    real = gemm.pair_linear
    monkeypatch.setattr(gemm, "pair_linear", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    losses, trace, model, g = _run_golden(name, "bf16x2f")
    assert bool(calls) == pair, "the pair forward %s" % ("did not run" if pair else "ran")
    for k, v in losses.items():
        ref = float(g["loss/" + k])
        assert abs(float(v.detach()) - ref) <= LOSS_RTOL * max(abs(ref), 1e-6), (k, float(v.detach()), ref)
    for k in g.files:
        if k.startswith(("pseudo_", "pgt_instance_")):
            np.testing.assert_array_equal(trace[k].cpu().numpy(), g[k], err_msg=k)
    for n, p in model.named_parameters():
        key = "gradnorm/" + n
        if key in g.files and float(g[key]) > 1e-5:
            ref = float(g[key])
            assert abs(p.grad.double().norm().item() - ref) / ref <= MIXED_GRAD_TOL, n


# observed deviation of the single-plane bf16 mode from the reference (profiles/r02/precision_deviation.json, re-measured
# with the halo-tile convolution, whose K walk -- channel block major -- re-associates the fp32 sums and so moves which
# near-threshold selections the bf16 rounding flips): worst loss 18 % / 2.2 % / 13 % / 4.6 % / 3.9 %, worst gradient norm
# 20 % / 3.5 % / 26 % / 5.2 % / 4.7 %, selection sets that differ 4 of 15 / 0 / 4 of 9 / 0 / 0 -- bounds below = those with
# ~1.5x head-room.  (The parity modes are unaffected: bf16x3 losses <= 8.4e-6, bf16x2 <= 1.2e-4, no set differs.)
BF16_BOUNDS = {"e2e_voc_2img": (0.26, 0.30, 6), "e2e_voc_1img": (0.07, 0.09, 1), "e2e_align_1img": (0.20, 0.38, 6),
               "e2e_r50_2img": (0.07, 0.08, 1), "e2e_coco_2img": (0.06, 0.07, 3)}


@pytest.mark.parametrize("name", E2E_ALL)
def test_bf16_mode_tracks_the_reference(name):
    """The throughput mode (one bf16 plane per operand: what bench.py times) on all four goldens.  bf16 operands cannot
    meet the 1e-3 bar and may legitimately flip a near-threshold selection (which moves that branch's loss pair); the
    deviations must stay inside what was observed and recorded."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    losses, trace, model, g = _run_golden(name, "bf16")
    loss_tol, grad_tol, flips = BF16_BOUNDS[name]
    report = {k: (round(float(v.detach()), 5), round(float(g["loss/" + k]), 5)) for k, v in losses.items()}
    print("BF16REPORT", name, report)
    for k, (got, ref) in report.items():
        assert np.isfinite(got) and abs(got - ref) <= loss_tol * max(abs(ref), 1e-4), (k, report)
    differ = 0
    for k in g.files:
        if k.startswith(("pseudo_", "pgt_instance_")):
            a, b = trace[k].cpu().numpy(), g[k]
            differ += int(a.shape != b.shape or not np.array_equal(a, b))
    assert differ <= flips, differ
    for n, p in model.named_parameters():
        key = "gradnorm/" + n
        if key in g.files and float(g[key]) > 1e-5:
            ref = float(g[key])
            assert abs(p.grad.double().norm().item() - ref) <= grad_tol * ref, (n, p.grad.double().norm().item(), ref)


def test_engine_step_fused_predictor_equals_unfused(monkeypatch):
    """engine.FlatSGD lays the 8 predictor heads out as one matrix (one GEMM, no torch.cat, bias in the epilogue,
    weight gradient written in place); the same two training steps with that layout switched off must give the
    same losses and the same updated parameters up to bf16-GEMM re-association."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from od_wscl_amd import engine
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("ODW_NO_TIMER", "1")
    out = {}
    for fuse in (True, False):
        monkeypatch.setenv("ODW_NO_PRED_FUSE", "0" if fuse else "1")
        cfg = bench.build_cfg(21)
        step, _ = engine.build_training_step(cfg, dev, dtype="bf16", world=1, seed=cfg.SEED, backend="hip")
        images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 224, 150, 21, dev)
        losses = []
        for it in range(2):
            l, _ = step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20) + (it << 12), device=dev))
            losses.append({k: float(v.detach()) for k, v in l.items()})
        torch.cuda.synchronize()
        opt = step.optimizer
        pred = {n: opt.flat_p[o:o + k].clone() for n, (o, k) in opt.slices.items() if "predictor" in n}
        mom = {n: opt.flat_m[o:o + k].clone() for n, (o, k) in opt.slices.items() if "predictor" in n}
        out[fuse] = (losses, pred, mom)
    for it, (a, b) in enumerate(zip(out[True][0], out[False][0])):
        for k in a:           # step 1: same weights, same draws; step 2 also carries the run-to-run noise of the atomics
            assert abs(a[k] - b[k]) <= (1e-4, 5e-2)[it] * max(abs(b[k]), 1e-3), (it, k, a[k], b[k])
    for n in out[True][1]:
        ma, mb = out[True][2][n], out[False][2][n]                 # momentum after 2 steps = the gradients themselves
        assert (ma - mb).abs().max().item() <= 2e-2 * mb.abs().max().item() + 1e-7, n
        assert torch.allclose(out[True][1][n], out[False][1][n], rtol=0, atol=1e-6), n


def test_row_sparse_clean_backward_equals_dense(monkeypatch):
    """The clean half of the stacked fc6/fc7 pass takes no part in backward; the few hundred proposal rows the
    contrastive loss references are re-evaluated with their original dropout draws (GEMM row_ids) and carry the
    gradient instead.  Same first-step losses and the same gradients as the dense backward (ODW_NO_SPARSE=1)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from od_wscl_amd import engine
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("ODW_NO_TIMER", "1")
    out = {}
    for sparse in (True, False):
        monkeypatch.setenv("ODW_NO_SPARSE", "0" if sparse else "1")
        cfg = bench.build_cfg(21)
        step, _ = engine.build_training_step(cfg, dev, dtype="bf16", world=1, seed=cfg.SEED, backend="hip")
        images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 224, 150, 21, dev)
        l, _ = step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=1 << 20, device=dev))
        torch.cuda.synchronize()
        opt = step.optimizer
        grads = {n: opt.flat_m[o:o + k].clone() for n, (o, k) in opt.slices.items()}     # momentum after step 1 = d
        out[sparse] = ({k: float(v.detach()) for k, v in l.items()}, grads)
    for k in out[True][0]:
        a, b = out[True][0][k], out[False][0][k]
        assert abs(a - b) <= 1e-3 * max(abs(b), 1e-3), (k, a, b)
    worst = 0.0
    for n in out[True][1]:
        ga, gb = out[True][1][n], out[False][1][n]
        rel = (ga - gb).abs().max().item() / (gb.abs().max().item() + 1e-6)     # floor: analytically-zero gradients
        worst = max(worst, rel)
        assert rel <= 3e-2, (n, rel)          # bf16 GEMMs over different row sets: re-association + bf16 rounding


@pytest.mark.parametrize("name", ["e2e_voc_2img", "e2e_coco_2img"])
def test_reevaluated_clean_rows_equal_the_reattached_ones_in_the_bench_mode(name, monkeypatch):
    """"bf16x2f", the shared clean + DropBlock fc6 forward: the clean rows the contrastive loss differentiates are re-attached
    from the stacked pass's outputs (default) or RE-EVALUATED from the gathered planes (ODW_RECOMPUTE_CLEAN=1: the channel-major
    hi plane for the backward, the cell-major planes for the forward product, with the dropout draws of their original rows).
    Same losses, same gradients -- on goldens whose contrastive loss is not zero."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = {}
    for recompute in (False, True):
        monkeypatch.setenv("ODW_RECOMPUTE_CLEAN", "1" if recompute else "0")
        losses, trace, model, g = _run_golden(name, "bf16x2f")
        out[recompute] = ({k: float(v.detach()) for k, v in losses.items()},
                          {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        del model
    assert out[True][0]["loss_sim"] > 1e-6
    for k in out[True][0]:
        a, b = out[True][0][k], out[False][0][k]
        assert abs(a - b) <= 1e-5 * max(abs(b), 1e-3), (k, a, b)
    for n in out[True][1]:
        ga, gb = out[True][1][n], out[False][1][n]
        # (the re-evaluated rows' forward values come from other launches -- other tiling, other fp32 summation order --, so
        # the single-plane bf16 backward rounds slightly different operands: per tensor, relative L2 within the mode's
        # gradient tolerance, tests/test_fullsize_gpu.py GRAD_L2_TOL["bf16x2f"])
        assert (ga - gb).double().norm().item() <= 2e-2 * gb.double().norm().item() + 1e-7, n


def test_engine_steps_over_changing_shapes(monkeypatch):
    """Consecutive training steps on images of different sizes, proposal counts and label sets (the reference trains
    multi-scale: INPUT.MIN_SIZE_TRAIN has six sizes): every per-step buffer, index upload, split-K plan and
    weight-gradient batch is sized per step; losses stay finite and parameters move."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from od_wscl_amd import engine, synthetic
    from od_wscl_amd.structures import BoxList, to_image_list
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("ODW_NO_TIMER", "1")
    cfg = bench.build_cfg(21)
    step, _ = engine.build_training_step(cfg, dev, dtype="bf16", world=1, seed=cfg.SEED, backend="hip")
    opt = step.optimizer
    p0 = opt.flat_p.clone()
    cases = [(224, 320, 180, [3]), (352, 256, 90, [1, 7, 12]), (224, 320, 180, [3]), (160, 192, 40, [20, 5])]
    for it, (h, w, p, labs) in enumerate(cases):
        img = torch.from_numpy(synthetic.make_image(11, it, h, w))[:, :h, :w]
        boxes = torch.from_numpy(synthetic.make_proposals(11, it, p, h, w, min_size=16))
        images = to_image_list([img], 32).to(dev)
        rois = [BoxList(boxes.to(dev), (w, h), "xyxy")]
        t = BoxList(torch.zeros((len(labs), 4), device=dev), (w, h), "xyxy")
        t.add_field("labels", torch.tensor(labs, device=dev))
        t.add_field("labels_host", labs)
        losses, accs = step(images, [t], rois, DeviceRand(5, first_stream=(1 << 20) + (it << 12), device=dev), iteration=it + 1)
        vals = [float(v.detach()) for v in losses.values()]
        assert all(np.isfinite(v) for v in vals), (it, losses)
    torch.cuda.synchronize()
    assert torch.isfinite(opt.flat_p).all() and not torch.equal(opt.flat_p, p0)


def test_engine_step_with_several_images_per_rank(monkeypatch):
    """The reference trains IMS_PER_BATCH images per process (8 on its single-GPU setup, README.md:99-100): a 3-image
    batch through the engine, with the index staging ring starting far too small so that it has to grow."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from od_wscl_amd import engine
    from od_wscl_amd.modeling.roi_heads.weak_head import loss_fused
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("ODW_NO_TIMER", "1")
    real_init = loss_fused._Staging.__init__
    monkeypatch.setattr(loss_fused._Staging, "__init__", lambda self, device, slots=8, width=64: real_init(self, device, slots, 64))
    cfg = bench.build_cfg(21)
    step, _ = engine.build_training_step(cfg, dev, dtype="bf16", world=1, seed=cfg.SEED)
    images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 224, 120, 21, dev, n_images=3)
    assert len(rois) == 3 and images.tensors.shape[0] == 3
    p0 = step.optimizer.flat_p.clone()
    for it in range(2):
        losses, accs = step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=(1 << 20) + (it << 12), device=dev))
        assert all(np.isfinite(float(v.detach())) for v in losses.values()), losses
    torch.cuda.synchronize()
    assert step.model.roi_heads.loss_evaluator._staging.width > 64
    assert torch.isfinite(step.optimizer.flat_p).all() and not torch.equal(step.optimizer.flat_p, p0)


def test_iter_size_accumulates_gradients_over_a_group(monkeypatch):
    """SOLVER.ITER_SIZE = 2 (config/defaults.py:459-461, engine/trainer.py:86,118-120): no parameter moves after the
    first iteration of a group; after the second the momentum buffer (first optimiser step: buf = g + wd * p) holds the
    SUM of the two iterations' gradients -- checked against two single-iteration steps from the same weights."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from od_wscl_amd import engine
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("ODW_NO_TIMER", "1")
    batches = [bench.synthetic_batch(1234, r, 224, 150, 21, dev) for r in (0, 1)]
    rands = lambda it: DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev)

    def build(iter_size):
        cfg = bench.build_cfg(21)
        cfg.merge_from_list(["SOLVER.ITER_SIZE", iter_size, "SOLVER.WEIGHT_DECAY", 0.0, "SOLVER.WEIGHT_DECAY_BIAS", 0.0])
        return engine.build_training_step(cfg, dev, dtype="bf16x2f", world=1, seed=cfg.SEED)[0]

    step = build(2)
    p0 = step.optimizer.flat_p.clone()
    step(*batches[0], rands(0), iteration=1)
    torch.cuda.synchronize()
    assert torch.equal(step.optimizer.flat_p, p0), "the optimiser stepped inside an ITER_SIZE group"
    step(*batches[1], rands(1), iteration=2)
    torch.cuda.synchronize()
    assert not torch.equal(step.optimizer.flat_p, p0)
    m_group = step.optimizer.flat_m.clone()
    del step
    singles = []
    for k in (0, 1):
        one = build(-1)
        one(*batches[k], rands(k), iteration=1)
        torch.cuda.synchronize()
        singles.append(one.optimizer.flat_m.clone())
        slices = dict(one.optimizer.slices)
        del one
    want = singles[0] + singles[1]
    for n, (o, k) in slices.items():
        a, b = m_group[o:o + k], want[o:o + k]
        # (absolute floor: the det_score bias gradient is analytically zero -- a softmax over the proposals is shift
        # invariant -- and what is left of it, ~1e-8, is the rounding of sums whose terms are ~1e-3)
        assert (a - b).abs().max().item() <= 2e-3 * b.abs().max().item() + 2e-8, n


def test_mixed_batch_with_an_unlabelled_image_gets_background_labels():
    """odw_od_assign_indexed_dev with a device-side pseudo-GT count of 0 (an image without a positive label inside a
    batch that has some): every proposal background with weight 0 and zero targets, like od_layer's early return
    (pseudo_label_generator.py:167-170) -- not an assignment against uninitialised list entries."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import _lib as L
    from od_wscl_amd import synthetic
    boxes = torch.from_numpy(synthetic.make_proposals(3, 0, 300, 200, 200)).cuda()
    idx = torch.full((64,), 7, dtype=torch.int32, device="cuda")
    cls = torch.full((64,), 5, dtype=torch.int32, device="cuda")
    sc = torch.full((64,), float("nan"), device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    lab = torch.full((300,), -1, dtype=torch.int64, device="cuda")
    w = torch.full((300,), -1.0, device="cuda")
    t = torch.full((300, 4), -1.0, device="cuda")
    L.check(L.lib().odw_od_assign_indexed_dev(L.ptr(boxes), 300, L.ptr(idx), L.ptr(cls), L.ptr(sc), L.ptr(n), 64, 0.5, 10.0, 10.0, 5.0,
                                              5.0, L.ptr(lab), L.ptr(w), L.ptr(t), L.stream()), "od_assign")
    assert int(lab.abs().sum()) == 0 and float(w.abs().sum()) == 0.0 and float(t.abs().sum()) == 0.0


@pytest.mark.parametrize("fuse", [True, False])
def test_early_dense_backward_equals_the_ordinary_backward(monkeypatch, fuse):
    """engine.build_training_step lets the fused loss run the backward of the dense losses (predictor, DropBlock half of the
    stacked fc7 / fc6 pass) BEFORE it waits for the discovery lists, and finishes from loss_sim and the gradient of the
    stacked operand (LossDict.finish_backward).  Same kernels on the same operands: the gradients of one step must
    equal those of the ordinary single backward (up to the summation order of the bias-gradient atomics), with the
    predictor laid out as one matrix and as eight heads behind a torch.cat (whose gradients autograd delivers)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from od_wscl_amd import engine
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("ODW_NO_TIMER", "1")
    monkeypatch.setenv("ODW_NO_PRED_FUSE", "0" if fuse else "1")
    grads, losses = {}, {}
    for early in (True, False):
        monkeypatch.setenv("ODW_NO_EARLY_BWD", "0" if early else "1")
        cfg = bench.build_cfg(21)
        step, _ = engine.build_training_step(cfg, dev, dtype="bf16x2f", world=1, seed=cfg.SEED, backend="hip")
        assert step.model.roi_heads.loss_evaluator.early_backward == early
        images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 224, 150, 21, dev)
        l, _ = step(images, targets, rois, DeviceRand(cfg.SEED, first_stream=1 << 20, device=dev))
        torch.cuda.synchronize()
        assert (getattr(l, "finish_backward", None) is not None) == early
        opt = step.optimizer
        grads[early] = {n: opt.flat_g[o:o + k].double().clone() for n, (o, k) in opt.slices.items()}
        losses[early] = {k: float(v.detach()) for k, v in l.items()}
        del step, opt
    assert losses[True] == losses[False]
    for n, g in grads[False].items():
        if n.endswith("det_score.bias"):
            continue        # a softmax over the proposals is shift invariant: this gradient is rounding noise around 0
        d = (grads[True][n] - g).abs().max().item()
        assert d <= 1e-6 * g.abs().max().item() + 1e-9, (n, d, g.abs().max().item())


def test_graph_cache_stays_bounded_over_many_shapes(monkeypatch):
    """The reference trains six MIN_SIZE_TRAIN values with free aspect ratios (configs/voc/voc07_contra_db_b8_lr0.01_mcg.yaml:
    33-34): 12 distinct padded shapes between 480 and 1200 px, cycled three times through the bench's step function.  At most
    ODW.GRAPH_CACHE shapes stay captured, the allocator's reserve stops growing once the cache is full (an unbounded cache
    keeps one ~1-4 GB activation pool per shape forever), the rest of the steps run eagerly, losses stay finite."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from od_wscl_amd import engine, synthetic
    from od_wscl_amd.structures import BoxList, to_image_list
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("ODW_NO_TIMER", "1")
    cfg = bench.build_cfg(21)
    cfg.merge_from_list(["ODW.GRAPH_CACHE", 3])
    step, _ = engine.build_training_step(cfg, dev, dtype="bf16x2f", world=1, seed=cfg.SEED)
    body = step.model.hip_body()
    assert body.use_graphs and body.graph_cache_size == 3
    shapes = [(480, 640), (576, 768), (688, 912), (864, 1152), (1000, 1200), (1200, 900), (640, 480), (768, 576), (912, 688),
              (1152, 864), (1184, 1000), (800, 1200)]
    batches = []
    for k, (h, w) in enumerate(shapes):
        img = torch.from_numpy(synthetic.make_image(41, k, h, w))[:, :h, :w]
        boxes = torch.from_numpy(synthetic.make_proposals(41, k, 300, h, w, min_size=20))
        t = BoxList(torch.zeros((1, 4), device=dev), (w, h), "xyxy")
        t.add_field("labels", torch.tensor([1 + k % 20], device=dev))
        t.add_field("labels_host", [1 + k % 20])
        batches.append((to_image_list([img], 32).to(dev), [t], [BoxList(boxes.to(dev), (w, h), "xyxy")]))
    reserved, it = [], 0
    for cycle in range(3):
        for images, targets, rois in batches:
            losses, _ = step(images, targets, rois, DeviceRand(41, first_stream=(1 << 20) + (it << 12), device=dev))
            it += 1
            assert all(np.isfinite(float(v.detach())) for v in losses.values()), (cycle, it, losses)
        torch.cuda.synchronize()
        reserved.append(torch.cuda.memory_reserved(dev))
    st = body.graph_stats
    print("GRAPHCACHE", st, [round(r / 1e9, 2) for r in reserved])
    assert len(body._graphs) <= 3 and st["captures"] <= 2 + st["replays"] // 8 + 1 and st["eager"] >= 24
    assert reserved[2] <= reserved[1] * 1.10 + (1 << 28), reserved       # no growth once every shape has been seen
    # the same shape again and again (the bench): captured on its second sighting, replayed from then on
    images, targets, rois = batches[0]
    before = st["replays"]
    for k in range(12):
        step(images, targets, rois, DeviceRand(41, first_stream=(1 << 20) + ((it + k) << 12), device=dev))
    torch.cuda.synchronize()
    assert st["replays"] - before >= 9, st


def test_iter_size_follows_the_reference_when_batches_are_skipped(monkeypatch):
    """SOLVER.ITER_SIZE = 2 with skipped batches (engine/trainer.py:80-82 `continue`s over a batch with an unlabelled
    image; the optimiser steps when iteration % iter_size == 0 and zeroes the gradients only then, :118-120):
      * iterations 1, 2 run (step), 3 is skipped, 4 runs: the step after 4 sees the gradient of batch 4 ALONE -- not added
        to the already-applied sum of the previous group;
      * iterations 1 runs, 2 is skipped (no step), 3 and 4 run: the step after 4 sees batches 1 + 3 + 4.
    The momentum buffer after the first optimiser step is the gradient sum itself (weight decay off)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from od_wscl_amd import engine
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("ODW_NO_TIMER", "1")
    batches = {k: bench.synthetic_batch(1234, k, 224, 150, 21, dev) for k in (1, 2, 3, 4)}
    rands = lambda it: DeviceRand(1234, first_stream=(1 << 20) + (it << 12), device=dev)

    def build(iter_size, momentum=0.9):
        cfg = bench.build_cfg(21)
        cfg.merge_from_list(["SOLVER.ITER_SIZE", iter_size, "SOLVER.WEIGHT_DECAY", 0.0, "SOLVER.WEIGHT_DECAY_BIAS", 0.0,
                             "SOLVER.BASE_LR", 0.0, "SOLVER.MOMENTUM", momentum])
        return engine.build_training_step(cfg, dev, dtype="bf16x2f", world=1, seed=cfg.SEED)[0]

    def single(k):                          # the gradient of batch k alone, at the (never changing: lr 0) initial weights
        one = build(-1)
        one(*batches[k], rands(k), iteration=1)
        torch.cuda.synchronize()
        g = one.optimizer.flat_m.clone()
        return g, dict(one.optimizer.slices)

    g = {}
    for k in (1, 3, 4):
        g[k], slices = single(k)

    def close(a, b, what):
        for n, (o, k) in slices.items():
            x, y = a[o:o + k], b[o:o + k]
            assert (x - y).abs().max().item() <= 2e-3 * y.abs().max().item() + 1e-9, (what, n)

    # case 1: the first iteration of a group is skipped.  momentum 0: the buffer after a step IS that step's gradient sum
    step = build(2, momentum=0.0)
    step(*batches[1], rands(1), iteration=1)
    step(*batches[2], rands(2), iteration=2)          # optimiser step (1 + 2)
    assert step.optimizer.grads_clean
    step(*batches[4], rands(4), iteration=4)          # iteration 3 skipped; 4 % 2 == 0: optimiser step
    torch.cuda.synchronize()
    assert step.optimizer.grads_clean and step.optimizer.sched_steps == 1      # (the skipped iteration 3 held the scheduler step)
    close(step.optimizer.flat_m, g[4], "batch 4 alone")
    del step
    # case 2: the last iteration of a group is skipped
    step = build(2, momentum=0.0)
    p0 = step.optimizer.flat_p.clone()
    step(*batches[1], rands(1), iteration=1)
    step(*batches[3], rands(3), iteration=3)          # iteration 2 skipped: no optimiser step, the sum keeps growing
    torch.cuda.synchronize()
    assert not step.optimizer.grads_clean and step.optimizer.first, "the optimiser stepped although iteration 2 never ran"
    step(*batches[4], rands(4), iteration=4)
    torch.cuda.synchronize()
    assert step.optimizer.grads_clean and step.optimizer.sched_steps == 2
    close(step.optimizer.flat_m, g[1] + g[3] + g[4], "batches 1 + 3 + 4")
    assert torch.equal(step.optimizer.flat_p, p0)     # lr 0


@pytest.mark.parametrize("exact_hints", [True, False])
@pytest.mark.parametrize("name", ["e2e_voc_2img", "e2e_voc_1img", "e2e_coco_2img"])
def test_device_resident_lists_equal_the_host_assembled_lists(name, exact_hints, monkeypatch):
    """Round 6: the contrastive branch with its control flow on the device (csrc/loss_lists.hip, weak_head/loss_device.py:
    no host read inside the step) against the host-list path of rounds 2-5 (ODW_HOST_LISTS=1: two blocking reads, numpy
    assembly) on the same goldens: the same losses, the same selections, the same SupCon inputs.
    exact_hints: the device path plans its GEMMs for the live extents (read back: a test hook) -- then EVERY weight gradient
    is the host-list path's bit for bit (bias gradients are atomic column sums: 1e-6).  Without it the plans are made for
    bucketed hints: a differently split product re-associates its fp32 sums (1e-7), and the single-plane bf16 backward
    turns any such perturbation into rounding flips that saturate at bf16's resolution within a few layers (relative L2
    sqrt(delta * 2^-8) per layer: the backbone's deepest layers end up 1-3e-3 apart, the noise floor of this mode against
    the fp32 oracle, MIXED_GRAD_TOL); the head's gradients, one rounding from the top, stay within 1e-3."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd.modeling.roi_heads.weak_head import loss_device
    monkeypatch.delenv("ODW_HOST_LISTS", raising=False)
    monkeypatch.setattr(loss_device, "_EXACT", exact_hints)
    l_dev, t_dev, m_dev, g = _run_golden(name, "bf16x2f")
    assert t_dev.get("device_lists"), "the device-resident path was not taken in bf16x2f"
    grads_dev = {n: p.grad.detach().clone() for n, p in m_dev.named_parameters() if p.grad is not None}
    del m_dev
    monkeypatch.setenv("ODW_HOST_LISTS", "1")
    l_host, t_host, m_host, _ = _run_golden(name, "bf16x2f")
    assert not t_host.get("device_lists")
    for k in l_host:
        a, b = float(l_dev[k].detach()), float(l_host[k].detach())
        assert abs(a - b) <= (0.0 if exact_hints else 2e-6 * max(abs(b), 1e-6)), (k, a, b)
    assert t_dev["supcon_n"] == t_host["supcon_n"]
    for k in t_host:
        if k.startswith(("pseudo_", "pgt_instance_", "sim_new_", "iou_samples_")):
            assert torch.equal(t_dev[k].cpu(), t_host[k].cpu()), k
    assert torch.equal(t_dev["supcon_weights"].cpu(), t_host["supcon_weights"].cpu())
    for n, p in m_host.named_parameters():
        if p.grad is None:
            assert n not in grads_dev, n
            continue
        ref = p.grad.double()
        err = (grads_dev[n].double() - ref).norm().item() / max(ref.norm().item(), 1e-12)
        if exact_hints:
            # (Sim_Net's second Linear: the device path takes its weight gradient as ONE product over [views | re-attached rows],
            # the host-list path as two accumulating ones -- the same terms, another association; nothing else reads it)
            loose = n.endswith("bias") or n.endswith("model_sim.mlp.2.weight")
            assert torch.equal(grads_dev[n], p.grad) if not loose else err <= 1e-6, (n, err)
        else:
            assert err <= (MIXED_GRAD_TOL if n.startswith("backbone") else 1e-3) or ref.norm().item() < 1e-7, (n, err)


def test_maximum_taken_in_the_input_gradient_product_changes_no_bit(monkeypatch):
    """Round 6: fc6's input-gradient GEMM leaves max |dX| for ROI pooling's fixed-point backward (csrc/gemm_bf16.hip:
    epi_absmax_commit) instead of the 200 MB pre-pass (ODW_ABSMAX_PREPASS=1): the same scale, so every gradient is the
    pre-pass path's bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd.modeling.backbone import fc_extractor
    from od_wscl_amd.modeling.roi_heads.weak_head import loss_device
    monkeypatch.delenv("ODW_HOST_LISTS", raising=False)
    monkeypatch.setattr(loss_device, "_EXACT", True)
    taken = []
    orig = fc_extractor._PoolStack.backward

    def spy(ctx, dx):
        taken.append(getattr(ctx.holder, "absmax", None) is not None)
        return orig(ctx, dx)
    monkeypatch.setattr(fc_extractor._PoolStack, "backward", staticmethod(spy))
    monkeypatch.delenv("ODW_ABSMAX_PREPASS", raising=False)
    l_new, _, m_new, _ = _run_golden("e2e_voc_2img", "bf16x2f")
    assert taken and all(taken), "the input-gradient product did not leave its maximum with the pooling node"
    grads_new = {n: p.grad.detach().clone() for n, p in m_new.named_parameters() if p.grad is not None}
    del m_new
    taken.clear()
    monkeypatch.setenv("ODW_ABSMAX_PREPASS", "1")
    l_old, _, m_old, _ = _run_golden("e2e_voc_2img", "bf16x2f")
    assert taken and not any(taken)
    for k in l_old:
        assert float(l_new[k].detach()) == float(l_old[k].detach()), k
    for n, p in m_old.named_parameters():
        if p.grad is not None:
            if n.endswith("bias"):          # (atomic column sums)
                assert (grads_new[n] - p.grad).abs().max().item() <= 1e-6 * max(p.grad.abs().max().item(), 1e-12), n
            else:
                assert torch.equal(grads_new[n], p.grad), n
