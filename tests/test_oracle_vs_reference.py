"""The CPU oracle against vectors produced by the reference itself (tests/golden/).

No GPU.  These tests pin the checker: if they pass, comparing the HIP path with
the oracle is comparing it with the reference."""
import os

import numpy as np
import pytest
import torch

from conftest import e2e_arch, e2e_classes, e2e_inputs, load_e2e, weights_for
from oracle import hotpath_ref as H
from oracle import native


def test_roi_align_forward_matches_reference_cpu_kernel(ops_golden):
    g = ops_golden
    for sr in (0, 2):
        for scale in (0.125, 0.25):
            out = native.roi_align_fwd(g["ra_feat"], g["ra_rois"], scale, 7, 7, sr)
            np.testing.assert_array_equal(out, g["ra_out_sr%d_s%g" % (sr, scale)])
    np.testing.assert_array_equal(native.roi_align_fwd(g["ra_feat"], g["ra_rois"], 0.125, 3, 5, 0), g["ra_out_3x5"])


def test_nms_wetectron_rule_matches_reference_cpu_kernel(ops_golden):
    g = ops_golden
    for thr in (0.1, 0.3, 0.5, 0.7):
        keep = native.nms_wt(g["nms_boxes"], g["nms_scores"], thr, use_ge=True)
        np.testing.assert_array_equal(keep, g["nms_keep_ge_%g" % thr])
    np.testing.assert_array_equal(native.nms_wt(g["nms_eq_boxes"], g["nms_eq_scores"], 0.5, True), g["nms_eq_keep_ge"])
    # the CUDA rule (nms.cu:60, strict >) keeps the box whose IoU equals the threshold
    assert list(native.nms_wt(g["nms_eq_boxes"], g["nms_eq_scores"], 0.5, False)) == [0, 1, 2]
    assert list(g["nms_eq_keep_ge"]) == [0, 2]


def test_nms_torchvision_semantics_known_answers(ops_golden):
    # torchvision is absent from the reference tree: known-answer vectors for its documented rule
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 5], [0, 0, 10, 10], [20, 20, 30, 30]], np.float32)
    s = np.array([0.5, 0.9, 0.5, 0.1], np.float32)
    assert list(native.nms_tv(b, s, 0.5)) == [1, 0, 3]        # IoU(1,0)=0.5 is NOT > 0.5; tie 0/2 -> lower index
    assert list(native.nms_tv(b, s, 0.49)) == [1, 3]
    assert list(native.nms_tv(b[:0], s[:0], 0.5)) == []
    g = ops_golden
    out = g["easy_nms_cluster"][native.nms_tv(g["iou_a"][g["easy_nms_cluster"]], g["easy_nms_scores"][g["easy_nms_cluster"]], 0.1)]
    np.testing.assert_array_equal(out, g["easy_nms_out"])


def test_box_iou_and_cal_iou(ops_golden):
    g = ops_golden
    np.testing.assert_array_equal(native.box_iou(g["iou_a"], g["iou_b"]), g["iou_ab"])
    a = torch.from_numpy(g["iou_a"])
    np.testing.assert_array_equal(H.boxlist_iou(a, torch.from_numpy(g["iou_b"])).numpy(), g["iou_ab"])
    iou = H.boxlist_iou(a, a[3].view(1, 4))
    idx = torch.nonzero(torch.ge(iou, 0.5).max(dim=1)[0]).view(-1)
    np.testing.assert_array_equal(idx.numpy(), g["cal_iou_idx"])


def test_encode_and_smooth_l1(ops_golden):
    g = ops_golden
    enc = H.box_encode(torch.from_numpy(g["iou_a"][:7]), torch.from_numpy(g["iou_b"]))
    np.testing.assert_array_equal(enc.numpy(), g["encode_out"])
    out = H.smooth_l1(torch.from_numpy(g["sl1_x"]), torch.from_numpy(g["sl1_t"]), 1.0)
    np.testing.assert_array_equal(out.numpy(), g["sl1_out"])


def test_dropblock_given_the_draw(ops_golden):
    g = ops_golden

    class Fixed(object):
        def __init__(self, u):
            self.u = u

        def uniform(self, shape):
            return torch.from_numpy(self.u)
    for bs in (1, 3):
        out = H.dropblock(torch.from_numpy(g["db_x"]), bs, 0.3, Fixed(g["db_u_bs%d" % bs]))
        np.testing.assert_array_equal(out.numpy(), g["db_out_bs%d" % bs])


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_supcon_v2_forward_and_gradient(ops_golden, name):
    g = ops_golden
    F_, y, w = g["sc_%s_F" % name], g["sc_%s_labels" % name], g["sc_%s_w" % name]
    loss, dF = native.supcon_v2(F_, y, w, 0.2)
    assert abs(loss - float(g["sc_%s_loss" % name])) <= 2e-6 * abs(float(g["sc_%s_loss" % name]))
    ref = g["sc_%s_dF" % name]
    assert np.abs(dF - ref).max() <= 2e-5 * np.abs(ref).max()
    # torch restatement (class-major features, weights in append order)
    ft = torch.from_numpy(F_.copy()).requires_grad_(True)
    enc = [ft[torch.from_numpy(y == c)] for c in range(int(y.max()) + 1)]
    l2 = H.supcon_v2(enc, torch.from_numpy(w), 0.2)[0]
    assert abs(l2.item() - float(g["sc_%s_loss" % name])) <= 1e-6 * abs(float(g["sc_%s_loss" % name]))


def test_od_layer(ops_golden):
    g = ops_golden
    pgt = [torch.zeros(0, dtype=torch.long) for _ in range(20)]
    pgt[3] = torch.from_numpy(g["od_pgt3"])
    pgt[10] = torch.from_numpy(g["od_pgt10"])
    pl, lw, rt = H.od_layer(torch.from_numpy(g["iou_a"]), torch.from_numpy(g["od_score"]),
                            torch.from_numpy(g["od_labvec"]), pgt)
    np.testing.assert_array_equal(pl.numpy(), g["od_pseudo"])
    np.testing.assert_array_equal(lw.numpy(), g["od_weights"])
    np.testing.assert_array_equal(rt.numpy(), g["od_targets"])


def test_roi_pool_known_answers():
    """ROIPool has no CPU implementation in the reference (csrc/ROIPool.h:23): hand-checked cases of
    ROIPool_cuda.cu's arithmetic (C round(), +1 extent, floor/ceil bins, first max, empty bin -> 0/-1)."""
    feat = np.arange(2 * 1 * 4 * 6, dtype=np.float32).reshape(2, 1, 4, 6)
    feat[1] = -feat[1]
    rois = np.array([[0, 0, 0, 5, 3],        # scale 1: whole map, 2x3 pooling -> 2x2 bins
                     [1, 0, 0, 5, 3],        # negative map: first max = top-left of each bin
                     [0, 2.5, 0.5, 2.5, 0.5],  # round(2.5)=3 (half away from zero), round(.5)=1 -> 1x1 roi
                     [0, 10, 10, 12, 12]], np.float32)  # outside -> empty bins
    out, arg = native.roi_pool_fwd(feat, rois, 1.0, 2, 3)
    np.testing.assert_array_equal(out[0, 0], [[7, 9, 11], [19, 21, 23]])
    np.testing.assert_array_equal(arg[0, 0], [[7, 9, 11], [19, 21, 23]])
    np.testing.assert_array_equal(arg[1, 0], [[0, 2, 4], [12, 14, 16]])
    np.testing.assert_array_equal(out[1, 0], -np.array([[24, 26, 28], [36, 38, 40]], np.float32))
    assert (out[2, 0] == feat[0, 0, 1, 3]).all() and (arg[2, 0] == 1 * 6 + 3).all()
    assert (out[3] == 0).all() and (arg[3] == -1).all()
    gin = native.roi_pool_bwd(np.ones_like(out), arg, rois, feat.shape, 2, 3)
    assert gin[0, 0, 1, 3] == 6 + 1 and gin[0, 0, 1, 1] == 1 and gin.sum() == 6 + 6 + 6


E2E = ["e2e_voc_2img", "e2e_voc_1img", "e2e_align_1img", "e2e_r50_2img", "e2e_coco_2img"]


@pytest.mark.parametrize("name", E2E)
def test_end_to_end_matches_imported_reference(name):
    """Full train-mode forward + backward of the restated hot path vs the reference's own run."""
    g = load_e2e(name)
    seed, batch, boxes, labels, cfg = e2e_inputs(g)
    arch = e2e_arch(g)
    frozen = H.FROZEN if arch == "vgg16" else H.FROZEN_RESNET
    param_names = set(n for n, _ in H.param_shapes(e2e_classes(g), arch))
    sd = {}
    for k, v in weights_for(arch, e2e_classes(g)).items():
        t = torch.from_numpy(v.copy())
        if k in param_names and not k.startswith(frozen):
            t.requires_grad_(True)
        sd[k] = t
    tr = {}
    rand = H.Rand(seed)
    losses, accs = H.forward(batch, boxes, labels, sd, rand, cfg, tr)
    assert rand.s.next == int(g["streams_used"])           # same number of random draws, same order
    for k, v in losses.items():
        ref = float(g["loss/" + k])
        assert abs(float(v) - ref) <= 1e-5 * max(abs(ref), 1e-6), (k, float(v), ref)
    for k, v in accs.items():
        assert abs(float(v) - float(g["acc/" + k])) < 1e-6
    for k in g.files:                                       # index selection must be bit-exact
        if k.startswith(("pseudo_", "pgt_instance_")):
            np.testing.assert_array_equal(tr[k].numpy(), g[k], err_msg=k)
        if k.startswith("weights_"):
            np.testing.assert_allclose(tr[k].numpy(), g[k], rtol=1e-5, atol=1e-9)
    assert int(tr["supcon_n"]) == int(g["supcon_n"])
    np.testing.assert_allclose(tr["supcon_weights"].numpy(), g["supcon_weights"], rtol=1e-5, atol=1e-12)
    sum(losses.values()).backward()
    for n, p in sd.items():
        key = "gradnorm/" + n
        if key in g.files:
            ref = float(g[key])
            assert abs(p.grad.double().norm().item() - ref) <= 1e-4 * max(ref, 1e-9), n
        else:
            assert p.grad is None or n.startswith(frozen)


@pytest.mark.skipif(not os.path.isdir("/root/reference/wetectron"), reason="reference tree not present")
def test_reference_build_exports_expected_functions():
    """oracle/_ref is the reference's own csrc/cpu compiled in place; it must export the functions
    the golden vectors were produced with and refuse ROIPool on CPU like the reference does."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import build_ref
    build_ref.build()
    m = build_ref.load_ref()
    assert m is not None
    for fn in ("nms", "roi_align_forward", "roi_pool_forward", "roi_align_backward"):
        assert hasattr(m, fn)
    with pytest.raises(RuntimeError):
        m.roi_pool_forward(torch.zeros(1, 1, 4, 4), torch.zeros(1, 5), 1.0, 2, 2)


def test_inference_tail_matches_imported_reference(weights_np):
    """Eval forward + PostProcessor of the restated path vs the detections the imported reference returned."""
    from oracle import inference_ref as I
    from od_wscl_amd import synthetic
    g = load_e2e("infer_voc_2img")
    seed, batch, boxes, _, cfg = e2e_inputs_infer(g)
    sd = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    with torch.no_grad():
        out = I.forward_eval(batch, boxes, [(int(w), int(h)) for h, w, _ in g["spec_images"]], sd,
                             dict(score_thresh=float(g["score_thresh"]), nms_test=float(g["nms"]), max_det=int(g["max_det"])))
    for i, (b, s, l) in enumerate(out):
        np.testing.assert_array_equal(l.numpy(), g["det_labels_%d" % i])
        np.testing.assert_allclose(b.numpy(), g["det_boxes_%d" % i], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(s.numpy(), g["det_scores_%d" % i], rtol=1e-6, atol=1e-8)


def tta_inputs(g):
    """pixels, proposals and the augmentation settings of a tests/golden/tta_*.npz case."""
    import voc_fixture
    from od_wscl_amd import synthetic
    seed = int(g["spec_seed"])
    specs = [(int(h), int(w), int(p)) for h, w, p in g["spec_images"]]
    pixels, _, _ = voc_fixture.make_case(seed, [(h, w) for h, w, _ in specs])
    boxes = [synthetic.make_proposals(seed, k, p, h, w, min_size=12) for k, (h, w, p) in enumerate(specs)]
    aug = dict(min_test=int(g["aug_min_test"]), max_test=int(g["aug_max_test"]), h_flip=bool(g["aug_h_flip"]),
               scales=tuple(int(v) for v in g["aug_scales"].tolist()), max_size=int(g["aug_max_size"]),
               scale_h_flip=bool(g["aug_scale_h_flip"]), mean=g["pixel_mean"], std=g["pixel_std"],
               to_bgr255=bool(g["to_bgr255"]), size_divisible=32, heur=str(g["heur"]) if "heur" in g.files else "AVG")
    return specs, pixels, boxes, aug


@pytest.mark.parametrize("case", ["tta_voc_2img", "tta_union_2img"])
def test_test_time_augmentation_matches_imported_reference(weights_np, case):
    """im_detect_bbox_aug (engine/bbox_aug.py) restated: 6 passes over a 2-image batch with the AVG merge, 3 passes
    with the UNION merge; filter."""
    from oracle import inference_ref as I
    g = load_e2e(case)
    specs, pixels, boxes, aug = tta_inputs(g)
    sd = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    with torch.no_grad():
        out = I.tta(pixels, boxes, sd, dict(score_thresh=float(g["score_thresh"]), nms_test=float(g["nms"]),
                                            max_det=int(g["max_det"])), aug)
    for i, (b, s, l) in enumerate(out):
        np.testing.assert_array_equal(l.numpy(), g["det_labels_%d" % i])
        np.testing.assert_allclose(b.numpy(), g["det_boxes_%d" % i], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(s.numpy(), g["det_scores_%d" % i], rtol=1e-6, atol=1e-8)


def e2e_inputs_infer(g):
    import torch as _t
    from od_wscl_amd import synthetic
    seed = int(g["spec_seed"])
    specs = g["spec_images"]
    hm = max(synthetic.pad_to(int(h)) for h, w, p in specs)
    wm = max(synthetic.pad_to(int(w)) for h, w, p in specs)
    batch = _t.zeros(len(specs), 3, hm, wm)
    boxes = []
    for k, (h, w, p) in enumerate(specs):
        h, w, p = int(h), int(w), int(p)
        batch[k, :, :h, :w] = _t.from_numpy(synthetic.make_image(seed, k, h, w)[:, :h, :w].copy())
        boxes.append(_t.from_numpy(synthetic.make_proposals(seed, k, p, h, w, min_size=12)))
    return seed, batch, boxes, None, {}


@pytest.mark.skipif(not os.path.isdir("/root/reference/wetectron"), reason="reference tree not present")
@pytest.mark.parametrize("yaml_rel,arch", [("configs/voc/voc07_contra_db_b8_lr0.01_mcg.yaml", "vgg16"),
                                           ("configs/voc/voc07_r50_c5_contra_db_b8_lr0.02_ss.yaml", "r50"),
                                           ("configs/voc/voc07_r101_c5_contra_db_b8_lr0.02_ss.yaml", "r101")])
def test_state_dict_layout_equals_the_reference_model(yaml_rel, arch):
    """Checkpoint compatibility: the imported reference model and ours, built from the same
    yaml, expose the same state-dict keys with the same shapes, so `.pth` files travel both ways."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import refimport
    from od_wscl_amd.config import cfg as base
    from od_wscl_amd.modeling.detector import build_detection_model
    from od_wscl_amd.utils import checkpoint as ck
    ref = refimport.build_reference_model(refimport.reference_cfg(yaml_rel))
    cfg = base.clone()
    cfg.merge_from_file(os.path.join("/root/reference", yaml_rel))
    ours = build_detection_model(cfg)
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    assert a == b
    matched = ck.load_state_dict(ours, {"module." + k: v for k, v in ref.state_dict().items()})
    assert len(matched) == len(a)
    for (k, v) in ours.state_dict().items():
        assert torch.equal(v, ref.state_dict()[k]), k


def test_oracle_generator_is_independent_of_and_identical_to_the_product_generator():
    """oracle/rng_ref.py restates the counter-based generator without importing the product's host twin; the two must
    produce the same bits (the device kernel is pinned against the product's in tests/test_e2e_gpu.py)."""
    import inspect
    from oracle import rng_ref
    from od_wscl_amd.utils import rng
    assert "od_wscl_amd" not in inspect.getsource(rng_ref).split('"""', 2)[2]
    for seed, stream, n, off in [(0, 0, 17, 0), (58, 3, 1000, 5), (2 ** 31 + 7, 2 ** 20 + 9, 4097, 123457), (1234, 77, 64, 2 ** 32 - 10)]:
        np.testing.assert_array_equal(rng_ref.bits(seed, stream, n, off), rng.bits(seed, stream, n, off))
        np.testing.assert_array_equal(rng_ref.uniform(seed, stream, n, off), rng.uniform(seed, stream, n, off))
        np.testing.assert_array_equal(rng_ref.normal(seed, stream, n, off), rng.normal(seed, stream, n, off))
    assert rng_ref.stream_key(58, 3) == rng.stream_key(58, 3)


def test_committed_goldens_are_what_the_reference_produces():
    """`python tests/golden/make_golden.py --check`: every golden file regenerated from the imported reference (and
    its compiled CPU operators) into a scratch directory equals the committed file bit for bit -- same keys, same
    arrays.  Runs only where /root/reference exists (this container); the GPU box sees the committed arrays only."""
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/wetectron"):
        pytest.skip("reference tree not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "make_golden.py"), "--check"],
                       capture_output=True, text=True, timeout=900)
    diffs = [l for l in r.stdout.splitlines() if l.startswith("DIFF")]
    assert r.returncode == 0 and not diffs, "\n".join(diffs[:40]) + "\n" + r.stderr[-2000:]
