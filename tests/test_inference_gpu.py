"""The inference tail on the GPU (csrc/detect.hip behind modeling/roi_heads/box_head/inference.py:PostProcessor):
against the CPU oracle on random scores / regressions, and the whole eval forward against the detections the
imported reference produced (tests/golden/infer_voc_2img.npz).  `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from conftest import load_e2e, weights_for
from test_oracle_vs_reference import e2e_inputs_infer, tta_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,C,thresh,nms,regress", [(300, 21, 0.0, 0.4, True), (1000, 21, 0.03, 0.3, True),
                                                    (257, 5, 0.0, 0.5, False), (64, 81, 0.01, 0.4, True)])
def test_postprocessor_matches_oracle(P, C, thresh, nms, regress):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import inference_ref as I
    from od_wscl_amd import synthetic
    from od_wscl_amd.modeling.box_coder import BoxCoder
    from od_wscl_amd.modeling.roi_heads.box_head.inference import PostProcessor
    from od_wscl_amd.structures import BoxList
    from od_wscl_amd.utils import rng
    W, H = 320, 240
    sizes = [P - P // 3, P // 3]
    boxes = [torch.from_numpy(synthetic.make_proposals(7, k, n, H, W, min_size=8)) for k, n in enumerate(sizes)]
    prob = torch.softmax(torch.from_numpy(rng.normal(7, 3, P * C).reshape(P, C)) * 2, dim=1)
    reg = torch.from_numpy(rng.normal(7, 4, P * 4 * C).reshape(P, 4 * C)) * 0.5
    reg[5, 6] = 50.0                                                 # exercises the exp clamp (bbox_xform_clip)
    pp = PostProcessor(thresh, nms, 100, BoxCoder((10.0, 10.0, 5.0, 5.0)), False, False, regression=regress)
    bl = [BoxList(b.cuda(), (W, H), "xyxy") for b in boxes]
    res = pp((prob.cuda(), reg.cuda()), bl, softmax_on=False) if regress else pp(prob.cuda(), bl)
    o = 0
    for i, b in enumerate(boxes):
        n = b.shape[0]
        dec = I.decode(reg[o:o + n], b) if regress else b.repeat(1, C)
        d = dec.reshape(-1, 4).clone()
        d[:, 0].clamp_(min=0, max=W - 1); d[:, 1].clamp_(min=0, max=H - 1)
        d[:, 2].clamp_(min=0, max=W - 1); d[:, 3].clamp_(min=0, max=H - 1)
        ob, os_, ol = I.filter_results(d.reshape(n, -1), prob[o:o + n], (W, H), thresh, nms, 100)
        r = res[i]
        np.testing.assert_array_equal(r.get_field("labels").cpu().numpy(), ol.numpy())
        np.testing.assert_allclose(r.get_field("scores").cpu().numpy(), os_.numpy(), rtol=0, atol=0)
        np.testing.assert_allclose(r.bbox.cpu().numpy(), ob.numpy(), rtol=2e-6, atol=2e-4)
        o += n


def test_eval_forward_matches_reference_detections():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from test_e2e_gpu import build_model
    from od_wscl_amd.structures import BoxList, to_image_list
    g = load_e2e("infer_voc_2img")
    seed, batch, boxes, _, _ = e2e_inputs_infer(g)
    model = build_model(str(g["spec_pooler"]), weights_for("vgg16"), "fused")
    model.roi_heads.strong_post_processor.score_thresh = float(g["score_thresh"])
    model.roi_heads.strong_post_processor.nms = float(g["nms"])
    model.eval()
    rois = [BoxList(boxes[k].cuda(), (int(w), int(h)), "xyxy") for k, (h, w, p) in enumerate(g["spec_images"])]
    with torch.no_grad():
        res = model(to_image_list(batch.cuda()), rois=rois)
    for i, r in enumerate(res):
        np.testing.assert_array_equal(r.get_field("labels").cpu().numpy(), g["det_labels_%d" % i])
        np.testing.assert_allclose(r.get_field("scores").cpu().numpy(), g["det_scores_%d" % i], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(r.bbox.cpu().numpy(), g["det_boxes_%d" % i], rtol=1e-4, atol=1e-3)


def test_decode_then_filter_equals_the_fused_tail():
    """The two halves test-time augmentation uses (odw_detect_decode, odw_detect_filter) against the oracle and against
    the fused single-launch tail."""
    from oracle import inference_ref as I
    from od_wscl_amd import synthetic
    from od_wscl_amd.modeling.box_coder import BoxCoder
    from od_wscl_amd.modeling.roi_heads.box_head.inference import PostProcessor
    from od_wscl_amd.structures import BoxList
    from od_wscl_amd.utils import rng
    W, H, P, C = 300, 200, 500, 21
    sizes = [300, 200]
    boxes = [torch.from_numpy(synthetic.make_proposals(9, k, n, H, W, min_size=8)) for k, n in enumerate(sizes)]
    prob = torch.softmax(torch.from_numpy(rng.normal(9, 3, P * C).reshape(P, C)) * 2, dim=1)
    reg = torch.from_numpy(rng.normal(9, 4, P * 4 * C).reshape(P, 4 * C)) * 0.5
    bl = [BoxList(b.cuda(), (W, H), "xyxy") for b in boxes]
    fused = PostProcessor(0.01, 0.4, 100, BoxCoder((10.0, 10.0, 5.0, 5.0)), False, False)
    aug = PostProcessor(0.01, 0.4, 100, BoxCoder((10.0, 10.0, 5.0, 5.0)), False, True)
    want = fused((prob.cuda(), reg.cuda()), bl, softmax_on=False)
    raw = aug((prob.cuda(), reg.cuda()), bl, softmax_on=False)
    o = 0
    for i, b in enumerate(boxes):
        n = b.shape[0]
        d = I.decode(reg[o:o + n], b).reshape(-1, 4).clone()
        d[:, 0].clamp_(min=0, max=W - 1); d[:, 1].clamp_(min=0, max=H - 1)
        d[:, 2].clamp_(min=0, max=W - 1); d[:, 3].clamp_(min=0, max=H - 1)
        assert raw[i].bbox.shape == (n * C, 4) and raw[i].get_field("scores").shape == (n * C,)
        np.testing.assert_allclose(raw[i].bbox.cpu().numpy(), d.numpy(), rtol=2e-6, atol=2e-4)
        np.testing.assert_array_equal(raw[i].get_field("scores").cpu().numpy(), prob[o:o + n].reshape(-1).numpy())
        got = aug.filter_results(raw[i], C)
        for f in ("labels", "scores", "proposal_index"):
            np.testing.assert_array_equal(got.get_field(f).cpu().numpy(), want[i].get_field(f).cpu().numpy())
        np.testing.assert_array_equal(got.bbox.cpu().numpy(), want[i].bbox.cpu().numpy())
        o += n


@pytest.mark.parametrize("case", ["tta_voc_2img", "tta_union_2img"])
def test_test_time_augmentation_matches_reference_detections(case):
    """im_detect_bbox_aug: 6 passes (identity, flip, two scales + flips) of a 2-image batch from uint8 pixels --
    GPU preprocessing, eval forward, decode, un-flip / resize, AVG merge (or 3 passes and the UNION merge), filter --
    against the imported reference."""
    from test_e2e_gpu import build_model
    from oracle import data_ref as D
    from od_wscl_amd import bbox_aug
    from od_wscl_amd.config import make_defaults
    from od_wscl_amd.structures import BoxList
    g = load_e2e(case)
    specs, pixels, boxes, aug = tta_inputs(g)
    model = build_model("ROIPool", weights_for("vgg16"), "fused")
    pp = model.roi_heads.strong_post_processor
    pp.score_thresh, pp.nms, pp.bbox_aug_enabled = float(g["score_thresh"]), float(g["nms"]), True
    model.eval()
    cfg = make_defaults()
    cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.NUM_CLASSES", 21, "MODEL.ROI_HEADS.SCORE_THRESH", float(g["score_thresh"]),
                         "MODEL.ROI_HEADS.NMS", float(g["nms"]), "TEST.BBOX_AUG.ENABLED", True, "TEST.BBOX_AUG.HEUR", aug["heur"],
                         "TEST.BBOX_AUG.H_FLIP", aug["h_flip"], "TEST.BBOX_AUG.SCALES", aug["scales"],
                         "TEST.BBOX_AUG.MAX_SIZE", aug["max_size"], "TEST.BBOX_AUG.SCALE_H_FLIP", aug["scale_h_flip"],
                         "INPUT.MIN_SIZE_TEST", aug["min_test"], "INPUT.MAX_SIZE_TEST", aug["max_test"],
                         "DATALOADER.SIZE_DIVISIBILITY", 32])
    rois = [BoxList(torch.from_numpy(b), (w, h), "xyxy") for b, (h, w, p) in zip(boxes, specs)]
    with torch.no_grad():
        res = bbox_aug.im_detect_bbox_aug(model, pixels, torch.device("cuda:0"), rois, cfg)
    for i, r in enumerate(res):
        oh, ow = D.get_size((specs[i][1], specs[i][0]), aug["min_test"], aug["max_test"])
        assert r.size == (ow, oh)            # detections live in the frame of the first (identity-scale) pass
        np.testing.assert_array_equal(r.get_field("labels").cpu().numpy(), g["det_labels_%d" % i])
        np.testing.assert_allclose(r.get_field("scores").cpu().numpy(), g["det_scores_%d" % i], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(r.bbox.cpu().numpy(), g["det_boxes_%d" % i], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("regress", [True, False])
def test_per_class_fallback_equals_the_fused_kernel(regress):
    """More proposals per image than the fused kernel holds in LDS (4096; the reference has no cap at test time) take
    the decode + per-class odw_nms path: forced here at P = 600 by lowering the switch, it must return exactly what the
    fused kernel returns -- and a real P = 5000 image must run."""
    from od_wscl_amd import synthetic
    from od_wscl_amd.modeling.box_coder import BoxCoder
    from od_wscl_amd.modeling.roi_heads.box_head.inference import PostProcessor
    from od_wscl_amd.structures import BoxList
    from od_wscl_amd.utils import rng
    W, H, C = 320, 240, 21
    for P in (600, 5000):
        sizes = [P - P // 3, P // 3] if P == 600 else [P]
        boxes = [torch.from_numpy(synthetic.make_proposals(9, k, n, H, W, min_size=4)) for k, n in enumerate(sizes)]
        prob = torch.softmax(torch.from_numpy(rng.normal(9, 3, P * C).reshape(P, C)) * 2, dim=1).cuda()
        reg = (torch.from_numpy(rng.normal(9, 4, P * 4 * C).reshape(P, 4 * C)) * 0.5).cuda()
        bl = [BoxList(b.cuda(), (W, H), "xyxy") for b in boxes]
        pp = PostProcessor(0.01, 0.4, 100, BoxCoder((10.0, 10.0, 5.0, 5.0)), False, False, regression=regress)
        x = (prob, reg) if regress else prob
        kw = dict(softmax_on=False) if regress else {}
        if P == 600:
            want = pp(x, bl, **kw)
            pp.FUSED_MAX_P = 100
        got = pp(x, bl, **kw)
        if P == 600:
            for a, b in zip(got, want):
                np.testing.assert_array_equal(a.get_field("labels").cpu().numpy(), b.get_field("labels").cpu().numpy())
                np.testing.assert_array_equal(a.get_field("scores").cpu().numpy(), b.get_field("scores").cpu().numpy())
                np.testing.assert_array_equal(a.bbox.cpu().numpy(), b.bbox.cpu().numpy())
                np.testing.assert_array_equal(a.get_field("proposal_index").cpu().numpy(),
                                              b.get_field("proposal_index").cpu().numpy())
        else:
            assert len(got) == 1 and 0 < len(got[0]) <= 120 and bool(torch.isfinite(got[0].bbox).all())
