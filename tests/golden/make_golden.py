"""Generate the committed golden vectors by running the REFERENCE in this container.

Usage (only where /root/reference exists):   python tests/golden/make_golden.py [names]
                                             python tests/golden/make_golden.py --check [names]
(--check regenerates into a scratch directory and compares with the committed files bit for bit; the CPU suite runs it:
tests/test_oracle_vs_reference.py::test_committed_goldens_are_what_the_reference_produces)

The reference Python is imported read-only through oracle/refimport.py (stub
packages for absent third-party modules, no reference code copied).  What is
stored is DATA: small inputs, or the seed/shape recipe of formula-generated
inputs, plus the reference's outputs.  The GPU box never sees the reference.

Files written next to this script:
  ops_ref.npz   per-operator vectors from the reference's own code:
                  - roi_align_forward  : compiled csrc/cpu/ROIAlign_cpu.cpp (oracle/_ref)
                  - nms (>= rule)      : compiled csrc/cpu/nms_cpu.cpp      (oracle/_ref)
                  - boxlist_iou, cal_iou, easy_nms, BoxCoder.encode, smooth_l1_loss,
                    DropBlock2D (given the uniform draw), od_layer, SupConLossV2 fwd + autograd
  e2e_*.npz     full train-mode forward+backward of build_detection_model(cfg) on
                formula-generated inputs: 8 losses, 4 accuracies, selected index sets,
                per-parameter gradient norms.  Randomness is injected (see _inject_rng).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import refimport  # noqa: E402
from od_wscl_amd import synthetic  # noqa: E402
from od_wscl_amd.utils import rng  # noqa: E402

E2E_CASES = {
    # name: (seed, [(H, W, P)], num_classes, pooler)
    "e2e_voc_2img": dict(seed=58, images=[(96, 128, 48), (80, 112, 40)], labels=[[3, 9], [9]], pooler="ROIPool"),
    "e2e_voc_1img": dict(seed=15, images=[(128, 128, 64)], labels=[[5]], pooler="ROIPool"),
    "e2e_align_1img": dict(seed=16, images=[(96, 96, 32)], labels=[[2, 17]], pooler="ROIAlign"),
    # config 5 of SURVEY.md s8: R-50-C5 body (stride 16) + ResNet50Conv5ROIFeatureExtractor
    "e2e_r50_2img": dict(seed=27, images=[(256, 320, 48), (224, 288, 40)], labels=[[7], [12]], pooler="ROIPool",
                         arch="r50", min_size=32, yaml="configs/voc/voc07_r50_c5_contra_db_b8_lr0.02_ss.yaml"),
    # config 4 of SURVEY.md s8: the COCO14 shape -- 81 classes (predictor N = 1377), labels up to 80
    # (one label per image: with 80 foreground classes and this weight set the same proposal tops both classes of a
    # two-label image in most seeds, which is quirk Q3's rounding-dependent `1.0 >= |e|^2` -- excluded by construction)
    "e2e_coco_2img": dict(seed=204, images=[(96, 128, 44), (80, 112, 36)], labels=[[17], [63]], pooler="ROIPool",
                          classes=81, yaml="configs/coco/coco14_contra_db_b8_lr0.01_mcg.yaml"),
}
# predictor / Sim_Net scales that give well separated scores (the reference's N(0,0.001)
# predictor init makes every score nearly tied, which no fp32 re-ordering survives)
OVERRIDES = {"predictor": 0.002, "model_sim.mlp.2": 0.05}
WEIGHT_SEED = 1   # one weight set for every e2e case (tests cache it)
CFG_OPTS = ["nms", 0.1, "lmda", 0.03, "temp", 0.2]


def edge_rois():
    """ROIs exercising rounding (x.5 after scaling), degenerate, out-of-bounds, full-image cases."""
    return np.array([
        [0, 0, 0, 95, 79], [0, 4, 4, 12, 12], [0, 12, 20, 52, 60], [1, 3.9, 4.1, 60.2, 70.7],
        [1, 20, 20, 20, 20], [0, 30, 30, 10, 10], [1, -40, -40, 20, 20], [0, 80, 60, 200, 200],
        [0, 2, 6, 10, 14], [1, 18, 22, 50, 58], [0, 90, 70, 95, 79], [1, 0, 0, 7, 7],
        [0, 1000, 1000, 1100, 1100], [1, 44, 36, 84, 76], [0, 6, 2, 90, 10], [1, 5, 5, 9, 75],
    ], np.float32)


def gen_ops(out):
    w = refimport.load_reference()
    ref_c = refimport._STATE["ref_c"]
    g = {}
    # ---- ROIAlign forward from the reference's compiled CPU kernel
    feat = rng.normal(3, 1, 2 * 5 * 10 * 12).reshape(2, 5, 10, 12)
    rois = edge_rois()
    extra = np.concatenate([np.zeros((24, 1), np.float32), synthetic.make_proposals(3, 0, 24, 80, 96, min_size=4)], 1)
    extra[::2, 0] = 1
    rois = np.concatenate([rois, extra], 0)
    g["ra_feat"], g["ra_rois"] = feat, rois
    for sr in (0, 2):
        for scale in (0.125, 0.25):
            o = ref_c.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois), scale, 7, 7, sr)
            g["ra_out_sr%d_s%g" % (sr, scale)] = o.numpy()
    o = ref_c.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois), 0.125, 3, 5, 0)
    g["ra_out_3x5"] = o.numpy()
    # ---- NMS from the reference's compiled CPU kernel (>= rule, +1 areas, ascending output)
    boxes = synthetic.make_proposals(5, 0, 300, 200, 300, min_size=8)
    scores = rng.uniform(5, 2, 300)
    scores[10:20] = scores[10]          # ties
    g["nms_boxes"], g["nms_scores"] = boxes, scores
    for thr in (0.1, 0.3, 0.5, 0.7):
        g["nms_keep_ge_%g" % thr] = ref_c.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy()
    # exact-threshold case: two boxes whose +1 IoU is exactly 0.5
    eb = np.array([[0, 0, 9, 9], [0, 0, 9, 4], [20, 20, 29, 29]], np.float32)
    es = np.array([0.9, 0.8, 0.7], np.float32)
    g["nms_eq_boxes"], g["nms_eq_scores"] = eb, es
    g["nms_eq_keep_ge"] = ref_c.nms(torch.from_numpy(eb), torch.from_numpy(es), 0.5).numpy()

    # ---- Python-side helpers of the reference
    from wetectron.structures.bounding_box import BoxList
    from wetectron.structures.boxlist_ops import boxlist_iou
    from wetectron.utils.utils import cal_iou, easy_nms
    from wetectron.modeling.box_coder import BoxCoder
    from wetectron.layers import smooth_l1_loss
    a = synthetic.make_proposals(6, 0, 40, 120, 160, min_size=8)
    b = synthetic.make_proposals(6, 1, 7, 120, 160, min_size=8)
    bl_a, bl_b = BoxList(torch.from_numpy(a), (160, 120), "xyxy"), BoxList(torch.from_numpy(b), (160, 120), "xyxy")
    g["iou_a"], g["iou_b"] = a, b
    g["iou_ab"] = boxlist_iou(bl_a, bl_b).numpy()
    idx, sc = cal_iou(bl_a, torch.tensor(3), 0.5)
    g["cal_iou_idx"], g["cal_iou_score"] = idx.numpy(), sc.numpy()
    cluster = torch.arange(0, 40, 2)
    csc = torch.from_numpy(rng.uniform(6, 5, 40))
    g["easy_nms_scores"] = csc.numpy()
    g["easy_nms_cluster"] = cluster.numpy()
    g["easy_nms_out"] = easy_nms(bl_a, cluster, csc, nms_iou=0.1).numpy()
    coder = BoxCoder(weights=(10.0, 10.0, 5.0, 5.0))
    g["encode_out"] = coder.encode(torch.from_numpy(a[:7]), torch.from_numpy(b)).numpy()
    x = torch.from_numpy(rng.normal(6, 7, 60).reshape(15, 4) * 2)
    t = torch.from_numpy(rng.normal(6, 8, 60).reshape(15, 4))
    g["sl1_x"], g["sl1_t"] = x.numpy(), t.numpy()
    g["sl1_out"] = smooth_l1_loss(x, t, beta=1, reduction=False).numpy()

    # ---- DropBlock2D given the uniform draw
    from wetectron.modeling.dropblock.drop_block import DropBlock2D
    xin = torch.from_numpy(rng.normal(8, 1, 6 * 4 * 7 * 7).reshape(6, 4, 7, 7))
    for bs in (1, 3):
        u = torch.from_numpy(rng.uniform(8, 10 + bs, 6 * 7 * 7).reshape(6, 7, 7))
        real_rand = torch.rand
        torch.rand = lambda *s, **k: u
        try:
            db = DropBlock2D(drop_prob=0.3, block_size=bs)
            db.train()
            g["db_out_bs%d" % bs] = db(xin).numpy()
        finally:
            torch.rand = real_rand
        g["db_u_bs%d" % bs] = u.numpy()
    g["db_x"] = xin.numpy()

    # ---- SupConLossV2 forward + autograd backward
    from wetectron.modeling.roi_heads.sim_head.sim_loss import SupConLossV2
    crit = SupConLossV2(0.2)
    for name, n, ncls in (("a", 7, 2), ("b", 64, 3), ("c", 300, 5), ("d", 97, 20)):
        Fm = rng.normal(9, ord(name), n * 128).reshape(n, 128)
        Fm /= np.linalg.norm(Fm, axis=1, keepdims=True)
        # class-major grouping like pgt_update: sizes differ per class, every class >= 2 rows
        cuts = np.sort((rng.uniform(9, 100 + ord(name), ncls - 1) * (n - 2 * ncls)).astype(np.int64)) + 2 * np.arange(1, ncls)
        cuts = np.concatenate([[0], cuts, [n]])
        ft = torch.from_numpy(Fm.copy()).requires_grad_(True)
        enc = [ft[cuts[c]:cuts[c + 1]] for c in range(ncls)]
        wts = torch.from_numpy(rng.uniform(9, 200 + ord(name), n))
        loss = crit(enc, wts, "cpu")
        loss.backward()
        labels = np.concatenate([np.full(cuts[c + 1] - cuts[c], c, np.int32) for c in range(ncls)])
        g["sc_%s_F" % name], g["sc_%s_labels" % name], g["sc_%s_w" % name] = Fm, labels, wts.numpy()
        g["sc_%s_loss" % name], g["sc_%s_dF" % name] = loss.item(), ft.grad.numpy()

    # ---- od_layer
    from wetectron.modeling.roi_heads.weak_head.pseudo_label_generator import od_layer
    refimport.reference_cfg(opts=CFG_OPTS)
    layer = od_layer()
    P, C = 40, 21
    score = torch.softmax(torch.from_numpy(rng.normal(10, 1, P * C).reshape(P, C) * 3), dim=1)
    labvec = torch.zeros(C)
    labvec[[4, 11]] = 1
    pgt = [torch.zeros(0, dtype=torch.long) for _ in range(C - 1)]
    pgt[3] = torch.tensor([5, 17, 2])
    pgt[10] = torch.tensor([int(torch.argmax(score[:, 11]))])
    pl, lw, rt = layer(bl_a, score, labvec, "cpu", pgt, return_targets=True)
    g["od_score"], g["od_labvec"] = score.numpy(), labvec.numpy()
    g["od_pgt3"], g["od_pgt10"] = pgt[3].numpy(), pgt[10].numpy()
    g["od_pseudo"], g["od_weights"], g["od_targets"] = pl.numpy(), lw.numpy(), rt.numpy()
    np.savez_compressed(out, **g)
    print("wrote", out, len(g), "arrays")


# --------------------------------------------------------------------------------- e2e
class _DetDropout(torch.nn.Module):
    def __init__(self, rand, p=0.5):
        super().__init__()
        self.rand, self.p = rand, p

    def forward(self, x):
        if not self.training:
            return x
        return self.rand.dropout(x, self.p)


def _inject_rng(model, rand):
    """Replace the reference's three random sources with the counter-based generator, consumed
    in the reference's own call order: nn.Dropout (vgg16.py:124,127), torch.rand
    (drop_block.py:42), torch.normal (vgg16.py:178)."""
    cl = model.roi_heads.feature_extractor.classifier
    for i, m in enumerate(cl):             # classifier.{3,6} for VGG16, .{2,5} for the ResNet extractor
        if isinstance(m, torch.nn.Dropout):
            cl[i] = _DetDropout(rand)
    real = (torch.rand, torch.normal)
    torch.rand = lambda *shape, **k: rand.uniform(tuple(shape))
    torch.normal = lambda mean, std, size=None, **k: rand.normal(tuple(size)) * std + mean
    return real


def gen_e2e(name, spec, out):
    from oracle import hotpath_ref as H
    refimport.load_reference()
    from wetectron.structures.bounding_box import BoxList
    from wetectron.structures.image_list import to_image_list
    arch = spec.get("arch", "vgg16")
    opts = CFG_OPTS + ["MODEL.ROI_BOX_HEAD.POOLER_METHOD", spec["pooler"]]
    cfg = refimport.reference_cfg(spec["yaml"], opts=opts) if "yaml" in spec else refimport.reference_cfg(opts=opts)
    model = refimport.build_reference_model(cfg)
    model.train()
    seed = spec["seed"]
    shapes = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    ncls = spec.get("classes", 21)
    assert shapes == H.param_shapes(ncls, arch), "parameter naming drifted"
    sd = synthetic.init_state_dict(shapes, WEIGHT_SEED, overrides=OVERRIDES)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(sd[n]))
    bufs = [(n, tuple(b.shape)) for n, b in model.named_buffers()]
    if arch != "vgg16":
        assert bufs == H.resnet_buffer_shapes(arch), "buffer naming drifted"
        bd = synthetic.init_buffers(bufs, WEIGHT_SEED)
        sd.update(bd)
        with torch.no_grad():
            for n, b in model.named_buffers():
                b.copy_(torch.from_numpy(bd[n]))
    imgs, rois, targets, boxes_np = [], [], [], []
    for k, (h, w, pcount) in enumerate(spec["images"]):
        imgs.append(torch.from_numpy(synthetic.make_image(seed, k, h, w)[:, :h, :w].copy()))
        bx = synthetic.make_proposals(seed, k, pcount, h, w, min_size=spec.get("min_size", 12))
        boxes_np.append(bx)
        rois.append(BoxList(torch.from_numpy(bx), (w, h), "xyxy"))
        t = BoxList(torch.zeros((len(spec["labels"][k]), 4)), (w, h), "xyxy")
        t.add_field("labels", torch.tensor(spec["labels"][k], dtype=torch.int64))
        targets.append(t)
    images = to_image_list(imgs, 32)
    rand = H.Rand(seed)
    real = _inject_rng(model, rand)
    rec = {}
    # record the reference's own intermediate selections
    import wetectron.modeling.roi_heads.weak_head.loss as L
    real_od = model.roi_heads.loss_evaluator.od_layer
    calls = []

    def od_spy(proposals, source_score, labels, device, pgt_instance, return_targets=False):
        r = real_od(proposals, source_score, labels, device, pgt_instance, return_targets=return_targets)
        calls.append((r[0].clone(), r[1].clone(), [p.clone() for p in pgt_instance]))
        return r
    model.roi_heads.loss_evaluator.od_layer = od_spy
    real_sim = model.roi_heads.loss_evaluator.sim_loss.forward

    def sim_spy(enc, col, device):
        rec["supcon_n"] = np.array(sum(e.shape[0] for e in enc))
        rec["supcon_weights"] = col.detach().numpy().copy()
        rec["supcon_class_sizes"] = np.array([e.shape[0] for e in enc])
        return real_sim(enc, col, device)
    model.roi_heads.loss_evaluator.sim_loss.forward = sim_spy
    try:
        losses, accs = model(images, targets, rois, iteration={"iter": 1})
        total = sum(losses.values())
        total.backward()
    finally:
        torch.rand, torch.normal = real
    n_ref = 3
    for idx in range(len(imgs)):
        for i in range(n_ref):
            pl, lw, pgt = calls[idx * n_ref + i]
            rec["pseudo_%d_%d" % (idx, i)] = pl.numpy()
            rec["weights_%d_%d" % (idx, i)] = lw.numpy()
            for c, p in enumerate(pgt):
                if p.numel():
                    rec["pgt_instance_%d_%d_%d" % (idx, i, c)] = p.numpy()
    for k, v in losses.items():
        rec["loss/" + k] = np.float64(v.item())
    for k, v in accs.items():
        rec["acc/" + k] = np.float64(float(v))
    for n, p in model.named_parameters():
        if p.grad is not None:
            rec["gradnorm/" + n] = np.float64(p.grad.double().norm().item())
            rec["gradsum/" + n] = np.float64(p.grad.double().sum().item())
    rec["spec_seed"] = np.array(seed)
    rec["spec_images"] = np.array(spec["images"])
    rec["spec_pooler"] = np.array(spec["pooler"])
    rec["spec_arch"] = np.array(arch)
    rec["spec_min_size"] = np.array(spec.get("min_size", 12))
    rec["spec_classes"] = np.array(ncls)
    rec["spec_labels_flat"] = np.array([l for ls in spec["labels"] for l in ls])
    rec["spec_labels_count"] = np.array([len(ls) for ls in spec["labels"]])
    rec["streams_used"] = np.array(rand.s.next)
    # decision margins (how far each data-dependent selection was from flipping), measured with
    # the oracle -- which reproduces the reference bit for bit on this machine.  Seeds are chosen
    # so that every margin is far above fp32 re-association noise; otherwise "bit-exact index
    # selection" would test the summation order of a K=25088 dot product, not the algorithm.
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    hm, wm = images.tensors.shape[-2:]
    tr = {}
    cfg_o = dict(nms=0.1, lmda=0.03, thres=0.5, temp=0.2, pooler=spec["pooler"], sampling_ratio=0, arch=arch,
                 scale=0.125 if arch == "vgg16" else 0.0625)
    with torch.no_grad():
        lo, _ = H.forward(images.tensors, [torch.from_numpy(b) for b in boxes_np],
                          [torch.tensor(l) for l in spec["labels"]], sdt, H.Rand(seed), cfg_o, tr)
    for k, v in lo.items():
        assert abs(float(v) - rec["loss/" + k]) <= 1e-6 * max(abs(rec["loss/" + k]), 1e-9), ("oracle != reference", k)
    floors = {"argmax_rel": 2e-3, "sim_thresh_abs": 2e-4, "q3_abs": 2e-4, "nms_order_rel": 2e-3}
    for k, v in tr.items():
        if k.startswith("margin/"):
            rec[k] = np.float64(float(v))
            print("   %-24s %.3e" % (k, float(v)))
            assert float(v) > floors[k[7:]], "fragile golden case: pick another seed (%s)" % k
    np.savez_compressed(out, **rec)
    print("wrote", out)
    for k in sorted(rec):
        if k.startswith(("loss/", "acc/")):
            print("   %-22s %.9g" % (k, rec[k]))
    print("   supcon N =", rec["supcon_n"], "class sizes", rec["supcon_class_sizes"][rec["supcon_class_sizes"] > 0])
    return rec


def gen_infer(name, spec, out):
    """Eval-mode forward of the imported reference (single scale: TEST.BBOX_AUG.ENABLED False): the detections
    PostProcessor returns for each image."""
    from oracle import hotpath_ref as H
    from oracle import inference_ref as I
    refimport.load_reference()
    from wetectron.structures.bounding_box import BoxList
    from wetectron.structures.image_list import to_image_list
    cfg = refimport.reference_cfg(opts=CFG_OPTS + ["MODEL.ROI_BOX_HEAD.POOLER_METHOD", spec["pooler"],
                                                    "TEST.BBOX_AUG.ENABLED", False])
    model = refimport.build_reference_model(cfg)
    model.eval()
    seed = spec["seed"]
    shapes = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    sd = synthetic.init_state_dict(shapes, WEIGHT_SEED, overrides=OVERRIDES)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(sd[n]))
    imgs, rois, boxes_np = [], [], []
    for k, (h, w, pcount) in enumerate(spec["images"]):
        imgs.append(torch.from_numpy(synthetic.make_image(seed, k, h, w)[:, :h, :w].copy()))
        bx = synthetic.make_proposals(seed, k, pcount, h, w, min_size=12)
        boxes_np.append(bx)
        rois.append(BoxList(torch.from_numpy(bx), (w, h), "xyxy"))
    images = to_image_list(imgs, 32)
    with torch.no_grad():
        result = model(images, rois=rois)
    rec = {"spec_seed": np.array(seed), "spec_images": np.array(spec["images"]), "spec_pooler": np.array(spec["pooler"]),
           "score_thresh": np.array(cfg.MODEL.ROI_HEADS.SCORE_THRESH), "nms": np.array(cfg.MODEL.ROI_HEADS.NMS),
           "max_det": np.array(cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG)}
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        ora = I.forward_eval(images.tensors, [torch.from_numpy(b) for b in boxes_np], [(w, h) for h, w, _ in spec["images"]], sdt,
                             dict(score_thresh=float(rec["score_thresh"]), nms_test=float(rec["nms"]), max_det=int(rec["max_det"])))
    for i, r in enumerate(result):
        rec["det_boxes_%d" % i] = r.bbox.numpy()
        rec["det_scores_%d" % i] = r.get_field("scores").numpy()
        rec["det_labels_%d" % i] = r.get_field("labels").numpy()
        ob, os_, ol = ora[i]
        assert np.array_equal(ol.numpy(), rec["det_labels_%d" % i]), "oracle != reference (labels)"
        np.testing.assert_allclose(ob.numpy(), rec["det_boxes_%d" % i], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(os_.numpy(), rec["det_scores_%d" % i], rtol=1e-6, atol=1e-8)
        print("   image %d: %d detections, classes %s" % (i, len(r), sorted(set(rec["det_labels_%d" % i].tolist()))[:8]))
    np.savez_compressed(out, **rec)
    print("wrote", out)


def gen_data(name, spec, out):
    """The imported reference's data boundary on a miniature VOC devkit: PascalVOCDataset.__getitem__ (proposal
    preparation, transforms) and BatchCollator, in training and in test mode, under fixed `random` / `torch` seeds."""
    import pickle
    import random
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    import voc_fixture
    from oracle import data_ref as D
    refimport.load_reference()
    from wetectron.data.transforms import build_transforms
    from wetectron.data.datasets.voc import PascalVOCDataset
    from wetectron.data.collate_batch import BatchCollator
    images, objects, proposals = voc_fixture.make_case(spec["seed"], spec["shapes"])
    ids = spec["ids"]
    rec = {"spec_seed": np.array(spec["seed"]), "spec_shapes": np.array(spec["shapes"]), "spec_ids": np.array(ids),
           "min_train": np.array(spec["min_train"]), "max_train": np.array(spec["max_train"]),
           "min_test": np.array(spec["min_test"]), "max_test": np.array(spec["max_test"])}
    cfg = refimport.reference_cfg(opts=["INPUT.MIN_SIZE_TRAIN", tuple(spec["min_train"]), "INPUT.MAX_SIZE_TRAIN",
                                        spec["max_train"], "INPUT.MIN_SIZE_TEST", spec["min_test"],
                                        "INPUT.MAX_SIZE_TEST", spec["max_test"], "DATALOADER.SIZE_DIVISIBILITY", 32])
    rec["pixel_mean"] = np.array(cfg.INPUT.PIXEL_MEAN, np.float32)
    rec["pixel_std"] = np.array(cfg.INPUT.PIXEL_STD, np.float32)
    rec["to_bgr255"] = np.array(cfg.INPUT.TO_BGR255)
    rec["pca"] = np.array(cfg.INPUT.PCA)
    with tempfile.TemporaryDirectory() as root:
        voc_fixture.write_devkit(root, "trainval", ids, images, objects)
        pkl = os.path.join(root, "props.pkl")
        with open(pkl, "wb") as f:
            pickle.dump(dict(boxes=proposals, scores=[np.ones(len(b), np.float32) for b in proposals],
                             indexes=[int(i) for i in ids]), f, pickle.HIGHEST_PROTOCOL)
        for mode, is_train in (("train", True), ("test", False)):
            ds = PascalVOCDataset(root, "trainval", use_difficult=not is_train,
                                  transforms=build_transforms(cfg, is_train), proposal_file=pkl)
            random.seed(spec["seed"])
            torch.manual_seed(spec["seed"])
            samples = [ds[i] for i in range(len(ids))]
            batch = BatchCollator(32)(samples)
            rec[mode + "_batch"] = batch[0].tensors.numpy()
            rec[mode + "_image_sizes"] = np.array([tuple(s) for s in batch[0].image_sizes])
            for i, (img, target, rois, index) in enumerate(samples):
                rec["%s_target_boxes_%d" % (mode, i)] = target.bbox.numpy()
                rec["%s_target_size_%d" % (mode, i)] = np.array(target.size)
                rec["%s_target_labels_%d" % (mode, i)] = target.get_field("labels").numpy()
                rec["%s_target_difficult_%d" % (mode, i)] = target.get_field("difficult").numpy()
                rec["%s_rois_%d" % (mode, i)] = rois.bbox.numpy()
                print("   %s image %d: %s -> %s, %d proposals of %d, %d objects" % (
                    mode, i, images[i].shape[:2], tuple(img.shape[1:]), len(rois), len(proposals[i]), len(target)))
        raw = PascalVOCDataset(root, "trainval", use_difficult=False, transforms=None, proposal_file=pkl)
        for i in range(len(ids)):
            _, _, rois, _ = raw[i]
            rec["raw_rois_%d" % i] = rois.bbox.numpy()
            np.testing.assert_array_equal(D.prepare_proposals(proposals[i], (images[i].shape[1], images[i].shape[0])),
                                          rec["raw_rois_%d" % i])
            info = raw.get_img_info(i)
            rec["info_%d" % i] = np.array([info["height"], info["width"]])
    np.savez_compressed(out, **rec)
    print("wrote", out)


def gen_tta(name, spec, out):
    """im_detect_bbox_aug of the imported reference (engine/bbox_aug.py): eval model with TEST.BBOX_AUG.ENABLED,
    "AVG" merge over the identity pass, its flip and two extra scales with flips."""
    from PIL import Image
    sys.path.insert(0, os.path.dirname(HERE))
    import voc_fixture
    from oracle import inference_ref as I
    refimport.load_reference()
    from wetectron.structures.bounding_box import BoxList
    from wetectron.engine.bbox_aug import im_detect_bbox_aug
    aug = spec["aug"]
    cfg = refimport.reference_cfg(opts=CFG_OPTS + [
        "MODEL.ROI_BOX_HEAD.POOLER_METHOD", "ROIPool", "TEST.BBOX_AUG.ENABLED", True, "TEST.BBOX_AUG.HEUR", spec.get("heur", "AVG"),
        "TEST.BBOX_AUG.H_FLIP", aug["h_flip"], "TEST.BBOX_AUG.SCALES", tuple(aug["scales"]),
        "TEST.BBOX_AUG.MAX_SIZE", aug["max_size"], "TEST.BBOX_AUG.SCALE_H_FLIP", aug["scale_h_flip"],
        "INPUT.MIN_SIZE_TEST", aug["min_test"], "INPUT.MAX_SIZE_TEST", aug["max_test"],
        "DATALOADER.SIZE_DIVISIBILITY", 32])
    model = refimport.build_reference_model(cfg)
    model.eval()
    shapes = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    sd = synthetic.init_state_dict(shapes, WEIGHT_SEED, overrides=OVERRIDES)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(sd[n]))
    pixels, _, _ = voc_fixture.make_case(spec["seed"], [(h, w) for h, w, _ in spec["images"]])
    rois, boxes_np = [], []
    for k, (h, w, pcount) in enumerate(spec["images"]):
        bx = synthetic.make_proposals(spec["seed"], k, pcount, h, w, min_size=12)
        boxes_np.append(bx)
        rois.append(BoxList(torch.from_numpy(bx), (w, h), "xyxy"))
    with torch.no_grad():
        result = im_detect_bbox_aug(model, [Image.fromarray(p, "RGB") for p in pixels], torch.device("cpu"), rois)
    rec = {"spec_seed": np.array(spec["seed"]), "spec_images": np.array(spec["images"]),
           "score_thresh": np.array(cfg.MODEL.ROI_HEADS.SCORE_THRESH), "nms": np.array(cfg.MODEL.ROI_HEADS.NMS),
           "max_det": np.array(cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG), "pixel_mean": np.array(cfg.INPUT.PIXEL_MEAN, np.float32),
           "pixel_std": np.array(cfg.INPUT.PIXEL_STD, np.float32), "to_bgr255": np.array(cfg.INPUT.TO_BGR255)}
    for k, v in aug.items():
        rec["aug_" + k] = np.array(v)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    ocfg = dict(score_thresh=float(rec["score_thresh"]), nms_test=float(rec["nms"]), max_det=int(rec["max_det"]))
    rec["heur"] = np.array(spec.get("heur", "AVG"))
    oaug = dict(aug, mean=rec["pixel_mean"], std=rec["pixel_std"], to_bgr255=bool(rec["to_bgr255"]), size_divisible=32,
                heur=spec.get("heur", "AVG"))
    with torch.no_grad():
        ora = I.tta(pixels, boxes_np, sdt, ocfg, oaug)
    for i, r in enumerate(result):
        rec["det_boxes_%d" % i] = r.bbox.numpy()
        rec["det_scores_%d" % i] = r.get_field("scores").numpy()
        rec["det_labels_%d" % i] = r.get_field("labels").numpy()
        ob, os_, ol = ora[i]
        assert np.array_equal(ol.numpy(), rec["det_labels_%d" % i]), "oracle != reference (labels)"
        np.testing.assert_allclose(ob.numpy(), rec["det_boxes_%d" % i], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(os_.numpy(), rec["det_scores_%d" % i], rtol=1e-6, atol=1e-8)
        print("   image %d: %d detections, classes %s" % (i, len(r), sorted(set(rec["det_labels_%d" % i].tolist()))[:8]))
    np.savez_compressed(out, **rec)
    print("wrote", out)


class _ListSampler(torch.utils.data.sampler.Sampler):
    def __init__(self, order):
        self.order = list(order)

    def __iter__(self):
        return iter(self.order)

    def __len__(self):
        return len(self.order)


def gen_sampler(name, spec, out):
    """The imported reference's GroupedBatchSampler: aspect-ratio grouping and the class-pair batches of
    SOLVER.CLASS_BATCH, on a two-split miniature VOC devkit under a fixed numpy seed."""
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    import voc_fixture
    refimport.load_reference()
    from wetectron.data.datasets import PascalVOCDataset, ConcatDataset
    from wetectron.data.samplers import GroupedBatchSampler, DistributedSampler, IterationBasedBatchSampler
    from wetectron.data.build import _quantize, _compute_aspect_ratios
    n = spec["n"]
    images, objects = voc_fixture.make_pair_case(spec["seed"], n)
    ids = ["%06d" % (k + 1) for k in range(n)]
    rec = {"spec_seed": np.array(spec["seed"]), "spec_n": np.array(n)}
    with tempfile.TemporaryDirectory() as root:
        half = n // 2
        voc_fixture.write_devkit(root, "train", ids[:half], images[:half], objects[:half])
        voc_fixture.write_devkit(root, "val", ids[half:], images[half:], objects[half:])
        args = [dict(data_dir=root, split="train", use_difficult=False, transforms=None),
                dict(data_dir=root, split="val", use_difficult=False, transforms=None)]
        ds = ConcatDataset([PascalVOCDataset(**a) for a in args])
        group_ids = _quantize(_compute_aspect_ratios(ds), [1])
        rec["group_ids"] = np.array(group_ids)
        order = np.random.RandomState(spec["seed"]).permutation(n).tolist()
        rec["order"] = np.array(order)
        for bs in (2, 3):
            b = list(GroupedBatchSampler(_ListSampler(order), group_ids, bs, bs * 2, ds, False, data_args=[args]))
            rec["grouped_bs%d" % bs] = np.array([x + [-1] * (bs - len(x)) for x in b])
        np.random.seed(spec["seed"])
        pairs = list(GroupedBatchSampler(_ListSampler(order), group_ids, 2, 4, ds, True, data_args=[args]))
        rec["class_pairs"] = np.array(pairs)
        rec["labels"] = np.array([(sorted(set(ds.datasets[ds.get_idxs(d)[0]].get_groundtruth(ds.get_idxs(d)[1])
                                             .get_field("labels").tolist())) + [-1, -1])[:2] for d in range(n)])
        for rank in range(2):
            s_ = DistributedSampler(ds, num_replicas=2, rank=rank, shuffle=True)
            s_.set_epoch(3)
            rec["dist_rank%d_epoch3" % rank] = np.array(list(s_))
        it = IterationBasedBatchSampler(GroupedBatchSampler(_ListSampler(order), group_ids, 2, 4, ds, False,
                                                            data_args=[args]), num_iterations=11, start_iter=4)
        rec["iteration_based"] = np.array([x + [-1] * (2 - len(x)) for x in it])
        print("   %d grouped batches, %d class pairs, %d iteration-based" % (len(b), len(pairs), len(rec["iteration_based"])))
    np.savez_compressed(out, **rec)
    print("wrote", out)


def gen_voc_eval(name, spec, out):
    """The imported reference's VOC metric (data/datasets/evaluation/voc/voc_eval.py) on random detections against a
    miniature devkit's annotations (difficult objects, duplicates, misses, empty images)."""
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    import voc_fixture
    refimport.load_reference()
    from wetectron.data.datasets import PascalVOCDataset
    from wetectron.structures.bounding_box import BoxList
    from wetectron.data.datasets.evaluation.voc.voc_eval import eval_detection_voc
    n = spec["n"]
    images, objects = voc_fixture.make_pair_case(spec["seed"], n)
    objects = [[(nm, int((k + j) % 4 == 0), x1, y1, x2, y2) for j, (nm, d, x1, y1, x2, y2) in enumerate(o)]
               for k, o in enumerate(objects)]
    ids = ["%06d" % (k + 1) for k in range(n)]
    preds = voc_fixture.make_detections(spec["seed"], images, objects)
    rec = {"spec_seed": np.array(spec["seed"]), "spec_n": np.array(n)}
    with tempfile.TemporaryDirectory() as root:
        voc_fixture.write_devkit(root, "test", ids, images, objects)
        ds = PascalVOCDataset(root, "test", use_difficult=True)
        pl, gl = [], []
        for k, (b, s, l) in enumerate(preds):
            info = ds.get_img_info(k)
            p = BoxList(torch.from_numpy(b), (info["width"], info["height"]), "xyxy")
            p.add_field("scores", torch.from_numpy(s))
            p.add_field("labels", torch.from_numpy(l))
            pl.append(p)
            gl.append(ds.get_groundtruth(k))
        for tag, m07 in (("07", True), ("area", False)):
            r = eval_detection_voc(pl, gl, iou_thresh=0.5, use_07_metric=m07)
            rec["ap_" + tag] = r["ap"]
            rec["map_" + tag] = np.array(r["map"])
            print("   %s: mAP %.4f" % (tag, r["map"]))
    np.savez_compressed(out, **rec)
    print("wrote", out)


def gen_coco(name, spec, out):
    """The imported reference's COCODataset (pycocotools / torchvision restated as index shims) on a miniature
    instances file: kept image ids, targets, proposals, get_groundtruth, with and without the annotation filter."""
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    import voc_fixture
    refimport.load_reference()
    from wetectron.data.datasets.coco import COCODataset
    data, pixels, proposals = voc_fixture.make_coco_case(spec["seed"], spec["n"])
    rec = {"spec_seed": np.array(spec["seed"]), "spec_n": np.array(spec["n"])}
    with tempfile.TemporaryDirectory() as root:
        ann, img_dir, pkl = voc_fixture.write_coco(root, data, pixels, proposals)
        for tag, remove in (("train", True), ("test", False)):
            ds = COCODataset(ann, img_dir, remove, transforms=None, proposal_file=pkl)
            rec[tag + "_ids"] = np.array(ds.ids)
            for i in range(len(ds)):
                img, target, rois, idx = ds[i]
                rec["%s_boxes_%d" % (tag, i)] = target.bbox.numpy()
                rec["%s_labels_%d" % (tag, i)] = target.get_field("labels").numpy()
                rec["%s_rois_%d" % (tag, i)] = rois.bbox.numpy()
                rec["%s_gt_%d" % (tag, i)] = ds.get_groundtruth(i).numpy()
                info = ds.get_img_info(i)
                rec["%s_info_%d" % (tag, i)] = np.array([info["height"], info["width"], info["id"]])
            print("   %s: %d of %d images kept" % (tag, len(ds), spec["n"]))
        rec["cat_map"] = np.array(sorted(ds.json_category_id_to_contiguous_id.items()))
    np.savez_compressed(out, **rec)
    print("wrote", out)


COCO_CASES = {"data_coco": dict(seed=17, n=8)}

EVAL_CASES = {"voc_eval": dict(seed=31, n=24)}

SAMPLER_CASES = {"sampler_voc": dict(seed=5, n=14)}

TTA_CASES = {"tta_voc_2img": dict(seed=23, images=[(96, 128, 48), (112, 80, 40)],
                                  aug=dict(min_test=96, max_test=160, h_flip=True, scales=(64, 128), max_size=176,
                                           scale_h_flip=True)),
             "tta_union_2img": dict(seed=29, images=[(96, 128, 40), (112, 80, 36)], heur="UNION",
                                    aug=dict(min_test=96, max_test=160, h_flip=True, scales=(128,), max_size=176,
                                             scale_h_flip=False))}

DATA_CASES = {"data_voc": dict(seed=11, shapes=[(60, 80), (75, 50), (64, 64), (48, 96)],
                               ids=["000005", "000007", "000009", "000012"], min_train=(48, 64, 80), max_train=100,
                               min_test=64, max_test=120)}

INFER_CASES = {"infer_voc_2img": dict(seed=58, images=[(96, 128, 48), (80, 112, 40)], pooler="ROIPool")}


def generate(which, out_dir):
    """Write the golden files named in `which` (all of them when empty) into `out_dir`; returns the paths written."""
    every = not which
    which = which or ["ops"] + list(E2E_CASES)
    written = []

    def run(fn, name, *args):
        path = os.path.join(out_dir, name + ".npz")
        fn(*args, path)
        written.append(path)

    if "ops" in which:
        run(gen_ops, "ops_ref")
    for name, spec in E2E_CASES.items():
        if name in which:
            run(gen_e2e, name, name, spec)
    for cases, fn in ((INFER_CASES, gen_infer), (COCO_CASES, gen_coco), (EVAL_CASES, gen_voc_eval), (SAMPLER_CASES, gen_sampler),
                      (TTA_CASES, gen_tta), (DATA_CASES, gen_data)):
        for name, spec in cases.items():
            if name in which or every:
                run(fn, name, name, spec)
    return written


def check(which):
    """Regenerate into a scratch directory and compare with the committed files: the same keys, and every array equal
    bit for bit (dtype, shape, content).  Returns the list of differences (empty = the committed vectors are what the
    reference produces from this script today)."""
    import tempfile
    problems = []
    with tempfile.TemporaryDirectory(prefix="odw_golden_") as tmp:
        for path in generate(which, tmp):
            name = os.path.basename(path)
            committed = os.path.join(HERE, name)
            if not os.path.exists(committed):
                problems.append("%s: not committed" % name)
                continue
            new, old = np.load(path, allow_pickle=False), np.load(committed, allow_pickle=False)
            for k in sorted(set(new.files) ^ set(old.files)):
                problems.append("%s: key %r only in the %s file" % (name, k, "regenerated" if k in new.files else "committed"))
            for k in sorted(set(new.files) & set(old.files)):
                a, b = new[k], old[k]
                if a.dtype != b.dtype or a.shape != b.shape:
                    problems.append("%s[%s]: %s%s regenerated, %s%s committed" % (name, k, a.dtype, a.shape, b.dtype, b.shape))
                elif a.tobytes() != b.tobytes():
                    if a.dtype.kind == "f":
                        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
                        problems.append("%s[%s]: %d of %d values differ (max abs %.3e)" % (name, k, int((d > 0).sum()), a.size, float(np.nanmax(d)) if a.size else 0.0))
                    else:
                        problems.append("%s[%s]: content differs" % (name, k))
    return problems


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--check":
        bad = check(args[1:])
        for line in bad:
            print("DIFF", line)
        print("golden check: %s" % ("%d difference(s)" % len(bad) if bad else "every regenerated file equals the committed one"))
        sys.exit(1 if bad else 0)
    generate(args, HERE)
