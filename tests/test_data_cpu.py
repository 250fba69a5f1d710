"""Data boundary on the CPU: the oracle against the golden vectors of the imported reference
and against the installed Pillow; the host half of the product (datasets, transforms' geometry and RNG consumption,
collation) against the same goldens.  The pixel half of the product only exists on the GPU (tests/test_data_gpu.py);
here its recorded plan is executed by the oracle."""
import os
import pickle
import random

import numpy as np
import pytest
import torch
from yacs_like import cfg_for_data  # noqa: E402  (tests/yacs_like.py)

import voc_fixture
from oracle import data_ref as D

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "data_voc.npz"))


def _devkit(tmp_path, g):
    shapes = [tuple(s) for s in g["spec_shapes"].tolist()]
    ids = [str(i) for i in g["spec_ids"].tolist()]
    images, objects, proposals = voc_fixture.make_case(int(g["spec_seed"]), shapes)
    root = str(tmp_path)
    voc_fixture.write_devkit(root, "trainval", ids, images, objects)
    from od_wscl_amd.data.datasets import ProposalFile
    pkl = os.path.join(root, "props.pkl")
    ProposalFile.write(pkl, proposals, [np.ones(len(b), np.float32) for b in proposals], [int(i) for i in ids])
    return root, pkl, images, proposals, ids


def test_resample_restatement_is_pillow_bit_for_bit():
    from PIL import Image
    rng = np.random.default_rng(0)
    for it in range(30):
        h, w = (int(v) for v in rng.integers(5, 90, 2))
        oh, ow = (int(v) for v in rng.integers(3, 140, 2))
        if it % 7 == 0:
            ow = w
        if it % 11 == 0:
            oh = h
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BILINEAR))
        np.testing.assert_array_equal(D.pil_bilinear_resize(img, oh, ow), ref)


def test_get_size_known_answers():
    # transforms.py:41-61: short side = size unless the long side would exceed max_size; truncating division
    assert D.get_size((500, 375), 600, 2000) == (600, 800)
    assert D.get_size((375, 500), 600, 2000) == (800, 600)
    assert D.get_size((500, 375), 1200, 1000) == (750, 1000)
    assert D.get_size((500, 333), 480, 2000) == (480, 720)
    assert D.get_size((333, 500), 333, 2000) == (500, 333)


def test_proposal_preparation_matches_the_reference(golden, tmp_path):
    root, pkl, images, proposals, ids = _devkit(tmp_path, golden)
    from od_wscl_amd.data.datasets import PascalVOCDataset, prepare_proposals
    ds = PascalVOCDataset(root, "trainval", use_difficult=False, transforms=None, proposal_file=pkl)
    assert len(ds) == len(ids)
    for i in range(len(ids)):
        size = (images[i].shape[1], images[i].shape[0])
        np.testing.assert_array_equal(D.prepare_proposals(proposals[i], size), golden["raw_rois_%d" % i])
        np.testing.assert_array_equal(prepare_proposals(proposals[i], size).bbox.numpy(), golden["raw_rois_%d" % i])
        img, target, rois, index = ds[i]
        assert index == i and img.size == size
        np.testing.assert_array_equal(rois.bbox.numpy(), golden["raw_rois_%d" % i])
        info = ds.get_img_info(i)
        assert [info["height"], info["width"]] == golden["info_%d" % i].tolist()


@pytest.mark.parametrize("mode", ["train", "test"])
def test_dataset_transforms_collate_match_the_reference(golden, tmp_path, mode):
    root, pkl, images, proposals, ids = _devkit(tmp_path, golden)
    from od_wscl_amd.data import BatchCollator, build_transforms
    from od_wscl_amd.data.datasets import PascalVOCDataset
    is_train = mode == "train"
    cfg = cfg_for_data(golden)
    ds = PascalVOCDataset(root, "trainval", use_difficult=not is_train, transforms=build_transforms(cfg, is_train),
                          proposal_file=pkl)
    random.seed(int(golden["spec_seed"]))
    torch.manual_seed(int(golden["spec_seed"]))
    samples = [ds[i] for i in range(len(ids))]
    pending, targets, rois, idx = BatchCollator(32)(samples)
    assert list(idx) == list(range(len(ids)))
    assert [tuple(s) for s in pending.image_sizes] == [tuple(s) for s in golden[mode + "_image_sizes"].tolist()]
    assert (len(pending), 3) + pending.padded_hw == golden[mode + "_batch"].shape
    for i in range(len(ids)):
        np.testing.assert_array_equal(targets[i].bbox.numpy(), golden["%s_target_boxes_%d" % (mode, i)])
        assert list(targets[i].size) == golden["%s_target_size_%d" % (mode, i)].tolist()
        np.testing.assert_array_equal(targets[i].get_field("labels").numpy(), golden["%s_target_labels_%d" % (mode, i)])
        np.testing.assert_array_equal(targets[i].get_field("difficult").numpy(),
                                      golden["%s_target_difficult_%d" % (mode, i)])
        np.testing.assert_array_equal(rois[i].bbox.numpy(), golden["%s_rois_%d" % (mode, i)])
    # the recorded pixel plans, executed by the oracle (Pillow and the written-out resampler), give the reference batch
    for use_pillow in (True, False):
        outs = []
        for im in pending.images:
            mean, std, bgr = im.norm
            outs.append(D.pixel_chain(im.pixels, im.shape[-2:], im.hflip, im.vflip, im.light, mean, std, bgr,
                                      use_pillow=use_pillow))
        batch, sizes = D.to_image_list(outs, 32)
        np.testing.assert_array_equal(batch, golden[mode + "_batch"])
    with pytest.raises(RuntimeError):
        pending.to("cpu")


def test_boxlist_geometry_known_answers():
    from od_wscl_amd.structures.bounding_box import BoxList, FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM
    b = BoxList(torch.tensor([[10.0, 20.0, 30.0, 50.0]]), (100, 80))
    assert b.transpose(FLIP_LEFT_RIGHT).bbox.tolist() == [[69.0, 20.0, 89.0, 50.0]]       # W - x - 1
    assert b.transpose(FLIP_TOP_BOTTOM).bbox.tolist() == [[10.0, 30.0, 30.0, 60.0]]       # H - y (no -1)
    assert b.resize((200, 160)).bbox.tolist() == [[20.0, 40.0, 60.0, 100.0]]
    assert b.resize((200, 80)).bbox.tolist() == [[20.0, 20.0, 60.0, 50.0]]
    np.testing.assert_array_equal(D.boxes_transpose(b.bbox.numpy(), (100, 80), 0), b.transpose(0).bbox.numpy())
    np.testing.assert_array_equal(D.boxes_resize(b.bbox.numpy(), (100, 80), (150, 90)), b.resize((150, 90)).bbox.numpy())
    c = BoxList(torch.tensor([[-5.0, 3.0, 120.0, 90.0], [4.0, 4.0, 4.0, 9.0]]), (100, 80)).clip_to_image()
    assert c.bbox.tolist() == [[0.0, 3.0, 99.0, 79.0]]


def test_live_reference_agrees_with_the_golden(golden, tmp_path):
    from oracle import refimport
    if not refimport.reference_available():
        pytest.skip("reference tree not present")
    refimport.load_reference()
    from wetectron.data.datasets.voc import PascalVOCDataset as RefVOC
    root, pkl, images, proposals, ids = _devkit(tmp_path, golden)
    ds = RefVOC(root, "trainval", use_difficult=False, transforms=None, proposal_file=pkl)
    for i in range(len(ids)):
        np.testing.assert_array_equal(ds[i][2].bbox.numpy(), golden["raw_rois_%d" % i])


class _ListSampler(torch.utils.data.sampler.Sampler):
    def __init__(self, order):
        self.order = list(order)

    def __iter__(self):
        return iter(self.order)

    def __len__(self):
        return len(self.order)


def test_samplers_match_the_reference(tmp_path):
    """Aspect-ratio grouping, SOLVER.CLASS_BATCH pairs, rank sharding and iteration-based resampling against what the
    imported reference's sampler classes produced on the same miniature two-split devkit."""
    g = np.load(os.path.join(HERE, "golden", "sampler_voc.npz"))
    n, seed = int(g["spec_n"]), int(g["spec_seed"])
    images, objects = voc_fixture.make_pair_case(seed, n)
    ids = ["%06d" % (k + 1) for k in range(n)]
    root = str(tmp_path)
    voc_fixture.write_devkit(root, "train", ids[: n // 2], images[: n // 2], objects[: n // 2])
    voc_fixture.write_devkit(root, "val", ids[n // 2:], images[n // 2:], objects[n // 2:])
    from od_wscl_amd.data import build as B
    from od_wscl_amd.data.datasets import ConcatDataset, PascalVOCDataset
    from od_wscl_amd.data.samplers import DistributedSampler, GroupedBatchSampler, IterationBasedBatchSampler
    ds = ConcatDataset([PascalVOCDataset(root, "train"), PascalVOCDataset(root, "val")])
    group_ids = B._quantize(B._compute_aspect_ratios(ds), [1])
    assert group_ids == g["group_ids"].tolist()
    order = g["order"].tolist()
    for bs in (2, 3):
        got = list(GroupedBatchSampler(_ListSampler(order), group_ids, bs))
        assert [x + [-1] * (bs - len(x)) for x in got] == g["grouped_bs%d" % bs].tolist()
    np.random.seed(seed)
    pairs = list(GroupedBatchSampler(_ListSampler(order), group_ids, 2, 4, ds, True))
    assert pairs == g["class_pairs"].tolist()
    labels = [set(ds.get_groundtruth(d).get_field("labels").tolist()) for d in range(n)]
    assert all(labels[a] & labels[b] for a, b in pairs)                   # every pair shares a class
    for rank in range(2):
        s = DistributedSampler(ds, num_replicas=2, rank=rank, shuffle=True)
        s.set_epoch(3)
        assert list(s) == g["dist_rank%d_epoch3" % rank].tolist()
    it = IterationBasedBatchSampler(GroupedBatchSampler(_ListSampler(order), group_ids, 2), num_iterations=11, start_iter=4)
    assert [x + [-1] * (2 - len(x)) for x in it] == g["iteration_based"].tolist() and len(it) == 11


def test_make_data_loader_end_to_end_on_a_devkit(golden, tmp_path):
    """cfg -> catalog -> datasets -> samplers -> workers -> collated pending batch (host half only)."""
    root, pkl, images, proposals, ids = _devkit(tmp_path, golden)
    from od_wscl_amd.data import make_data_loader

    class Catalog(object):
        @staticmethod
        def get(name):
            return dict(factory="PascalVOCDataset", args=dict(data_dir=root, split="trainval"))

    cfg = cfg_for_data(golden)
    cfg.merge_from_list(["DATASETS.TRAIN", ("voc_2007_trainval",), "DATASETS.TEST", ("voc_2007_trainval",),
                         "PROPOSAL_FILES.TRAIN", (pkl,), "PROPOSAL_FILES.TEST", (pkl,), "SOLVER.IMS_PER_BATCH", 2,
                         "SOLVER.MAX_ITER", 5, "TEST.IMS_PER_BATCH", 2, "DATALOADER.NUM_WORKERS", 2,
                         "DATALOADER.SIZE_DIVISIBILITY", 32])
    loader = make_data_loader(cfg, is_train=True, dataset_catalog=Catalog, num_gpus=1)
    batches = list(loader)
    assert len(batches) == 5
    for pending, targets, rois, idx in batches:
        assert 1 <= len(pending) <= 2 and len(targets) == len(rois) == len(idx) == len(pending)
        assert pending.padded_hw[0] % 32 == 0 and pending.padded_hw[1] % 32 == 0
        for im, r in zip(pending.images, rois):
            assert im.pixels.dtype == np.uint8 and r.size == im.size and im.norm is not None
    test_loaders = make_data_loader(cfg, is_train=False, dataset_catalog=Catalog, num_gpus=1)
    assert len(test_loaders) == 1 and sum(len(b[0]) for b in test_loaders[0]) == len(ids)
    cfg.merge_from_list(["TEST.BBOX_AUG.ENABLED", True])
    raw = next(iter(make_data_loader(cfg, is_train=False, dataset_catalog=Catalog, num_gpus=1)[0]))
    assert hasattr(raw[0][0], "size") and raw[2][0].size == raw[0][0].size       # PIL images + proposals, untransformed


def test_voc_metric_matches_the_reference(tmp_path):
    """mAP / per-class AP (VOC07 11-point and area rule) on random detections == the imported reference's
    eval_detection_voc on the same inputs; do_voc_evaluation end to end."""
    from od_wscl_amd.data.datasets import PascalVOCDataset
    from od_wscl_amd.data.evaluation import do_voc_evaluation, eval_detection_voc
    from od_wscl_amd.structures.bounding_box import BoxList
    g = np.load(os.path.join(HERE, "golden", "voc_eval.npz"))
    n, seed = int(g["spec_n"]), int(g["spec_seed"])
    images, objects = voc_fixture.make_pair_case(seed, n)
    objects = [[(nm, int((k + j) % 4 == 0), x1, y1, x2, y2) for j, (nm, d, x1, y1, x2, y2) in enumerate(o)]
               for k, o in enumerate(objects)]
    ids = ["%06d" % (k + 1) for k in range(n)]
    voc_fixture.write_devkit(str(tmp_path), "test", ids, images, objects)
    ds = PascalVOCDataset(str(tmp_path), "test", use_difficult=True)
    preds, gts = [], []
    for k, (b, s, l) in enumerate(voc_fixture.make_detections(seed, images, objects)):
        info = ds.get_img_info(k)
        p = BoxList(torch.from_numpy(b), (info["width"], info["height"]), "xyxy")
        p.add_field("scores", torch.from_numpy(s))
        p.add_field("labels", torch.from_numpy(l))
        preds.append(p)
        gts.append(ds.get_groundtruth(k))
    for tag, m07 in (("07", True), ("area", False)):
        r = eval_detection_voc(preds, gts, iou_thresh=0.5, use_07_metric=m07)
        np.testing.assert_allclose(r["ap"], g["ap_" + tag], rtol=0, atol=1e-12, equal_nan=True)
        assert abs(r["map"] - float(g["map_" + tag])) < 1e-12
    out = do_voc_evaluation(ds, preds, str(tmp_path))
    assert abs(out["map"] - float(g["map_07"])) < 1e-12 and os.path.exists(os.path.join(str(tmp_path), "result.txt"))


def test_coco_dataset_matches_the_reference(tmp_path):
    """COCODataset on the instances json (no pycocotools): kept ids, targets, proposals, class lists, image infos ==
    the imported reference's class on the same miniature file, with and without the annotation filter."""
    from od_wscl_amd.data.datasets import COCODataset
    g = np.load(os.path.join(HERE, "golden", "data_coco.npz"))
    data, pixels, proposals = voc_fixture.make_coco_case(int(g["spec_seed"]), int(g["spec_n"]))
    ann, img_dir, pkl = voc_fixture.write_coco(str(tmp_path), data, pixels, proposals)
    for tag, remove in (("train", True), ("test", False)):
        ds = COCODataset(ann, img_dir, remove, transforms=None, proposal_file=pkl)
        assert ds.ids == g[tag + "_ids"].tolist()
        for i in range(len(ds)):
            img, target, rois, idx = ds[i]
            assert idx == i
            np.testing.assert_array_equal(target.bbox.numpy(), g["%s_boxes_%d" % (tag, i)])
            np.testing.assert_array_equal(target.get_field("labels").numpy(), g["%s_labels_%d" % (tag, i)])
            np.testing.assert_array_equal(rois.bbox.numpy(), g["%s_rois_%d" % (tag, i)])
            np.testing.assert_array_equal(ds.get_groundtruth(i).numpy(), g["%s_gt_%d" % (tag, i)])
            info = ds.get_img_info(i)
            assert [info["height"], info["width"], info["id"]] == g["%s_info_%d" % (tag, i)].tolist()
    assert sorted(ds.json_category_id_to_contiguous_id.items()) == [tuple(r) for r in g["cat_map"].tolist()]
    # transforms + collation run on COCO samples like on VOC ones
    from od_wscl_amd.data import BatchCollator, build_transforms
    from od_wscl_amd.config import make_defaults
    cfg = make_defaults()
    cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", (48,), "INPUT.MAX_SIZE_TRAIN", 80])
    ds = COCODataset(ann, img_dir, True, transforms=build_transforms(cfg, True), proposal_file=pkl)
    pending, targets, rois, idx = BatchCollator(32)([ds[0], ds[1]])
    assert len(pending) == 2 and all(t.size == im.size for t, im in zip(targets, pending.images))
