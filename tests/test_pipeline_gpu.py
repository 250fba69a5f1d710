"""Data boundary -> hot path -> inference tail, end to end on the MI355X: a miniature VOC devkit goes through
make_data_loader (workers: decode + plans; GPU: preprocessing), the training step (forward, losses, backward, SGD),
the evaluation loop (single scale and test-time augmentation) and the VOC metric."""
import os
import sys

import numpy as np
import pytest
import torch

import voc_fixture

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(root, pkl):
    from od_wscl_amd.config import make_defaults
    cfg = make_defaults()
    cfg.merge_from_list(["MODEL.WSOD_ON", True, "MODEL.FASTER_RCNN", False, "MODEL.BACKBONE.CONV_BODY", "VGG16-OICR",
                         "MODEL.ROI_BOX_HEAD.NUM_CLASSES", 21, "MODEL.ROI_BOX_HEAD.POOLER_METHOD", "ROIPool",
                         "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", 7, "MODEL.ROI_BOX_HEAD.POOLER_SCALES", (0.125,),
                         "MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR", "VGG16.roi_head",
                         "MODEL.ROI_WEAK_HEAD.PREDICTOR", "MISTPredictor", "MODEL.ROI_WEAK_HEAD.LOSS", "RoIRegLoss",
                         "MODEL.ROI_WEAK_HEAD.OICR_P", 0.0, "MODEL.ROI_WEAK_HEAD.REGRESS_ON", True,
                         "MODEL.ROI_HEADS.SCORE_THRESH", 0.0, "MODEL.ROI_HEADS.NMS", 0.4,
                         "DB.METHOD", "dropblock", "SOLVER.CONTRA", True, "SOLVER.BASE_LR", 1e-5,
                         "SOLVER.IMS_PER_BATCH", 2, "SOLVER.MAX_ITER", 3, "TEST.IMS_PER_BATCH", 2,
                         "INPUT.MIN_SIZE_TRAIN", (160, 192), "INPUT.MAX_SIZE_TRAIN", 320, "INPUT.MIN_SIZE_TEST", 160,
                         "INPUT.MAX_SIZE_TEST", 320, "DATALOADER.SIZE_DIVISIBILITY", 32, "DATALOADER.NUM_WORKERS", 2,
                         "DATASETS.TRAIN", ("voc_2007_trainval",), "DATASETS.TEST", ("voc_2007_trainval",),
                         "PROPOSAL_FILES.TRAIN", (pkl,), "PROPOSAL_FILES.TEST", (pkl,), "SEED", 7])
    return cfg


def _devkit(tmp_path):
    from od_wscl_amd.data.datasets import ProposalFile
    shapes = [(120, 160), (150, 100), (128, 128), (96, 192)]
    ids = ["000005", "000007", "000009", "000012"]
    images, objects, _ = voc_fixture.make_case(3, shapes)
    rng = np.random.RandomState(3)
    proposals = []
    for h, w in shapes:                                   # 60 proposals per image, sides >= 20 px
        x1, y1 = rng.randint(0, w - 40, size=60), rng.randint(0, h - 40, size=60)
        proposals.append(np.stack([x1, y1, x1 + rng.randint(24, 40, size=60), y1 + rng.randint(24, 40, size=60)], 1))
    root = str(tmp_path)
    voc_fixture.write_devkit(root, "trainval", ids, images, objects)
    pkl = os.path.join(root, "props.pkl")
    ProposalFile.write(pkl, proposals, [np.ones(60, np.float32)] * 4, [int(i) for i in ids])

    class Catalog(object):
        @staticmethod
        def get(name):
            return dict(factory="PascalVOCDataset", args=dict(data_dir=root, split="trainval"))
    return root, pkl, Catalog, ids


def test_devkit_to_losses_to_map(tmp_path):
    from od_wscl_amd import engine, inference
    from od_wscl_amd.data import make_data_loader
    from od_wscl_amd.utils.device_rand import DeviceRand
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import test_net
    root, pkl, Catalog, ids = _devkit(tmp_path)
    cfg = _cfg(root, pkl)
    dev = torch.device("cuda:0")
    step, info = engine.build_training_step(cfg, dev, dtype="bf16", world=1, seed=7, backend="hip")
    loader = make_data_loader(cfg, is_train=True, dataset_catalog=Catalog, num_gpus=1)
    seen = 0
    for iteration, (images, targets, rois, _) in enumerate(loader, 1):
        for t in targets:
            t.add_field("labels_host", t.get_field("labels").tolist())
        batch = images.to(dev)
        assert batch.tensors.is_cuda and batch.tensors.shape[0] == len(rois)
        losses, accs = step(batch, [t.to(dev) for t in targets], [r.to(dev) for r in rois],
                            DeviceRand(7, first_stream=iteration << 12, device=dev), iteration=iteration)
        assert all(bool(torch.isfinite(v.detach()).all()) for v in losses.values()), losses
        seen += 1
    assert seen == 3
    # evaluation: single scale, then test-time augmentation, through the evaluation loop and the VOC metric
    model = test_net.build_eval_model(cfg, dev, "bf16")
    engine.load_formula_weights(model, 1)
    for aug in (False, True):
        cfg.merge_from_list(["TEST.BBOX_AUG.ENABLED", aug, "TEST.BBOX_AUG.HEUR", "AVG", "TEST.BBOX_AUG.H_FLIP", True,
                             "TEST.BBOX_AUG.SCALES", (128, 192), "TEST.BBOX_AUG.MAX_SIZE", 320,
                             "TEST.BBOX_AUG.SCALE_H_FLIP", True])
        model.roi_heads.strong_post_processor.bbox_aug_enabled = aug
        test_loader = make_data_loader(cfg, is_train=False, dataset_catalog=Catalog, num_gpus=1)[0]
        out = os.path.join(root, "eval_aug%d" % aug)
        os.makedirs(out)
        result = inference.inference(model, test_loader, "voc_2007_trainval", cfg, device=dev, output_folder=out)
        assert 0.0 <= result["map"] <= 1.0 and len(result["ap"]) >= 2
        preds = torch.load(os.path.join(out, "predictions.pth"), weights_only=False)
        assert len(preds) == len(ids) and all(len(p) > 0 and p.has_field("labels") for p in preds)


def test_train_net_checkpoints_and_resumes_with_its_momenta(tmp_path):
    """tools/train_net.py end to end (synthetic batches): periodic checkpoints in the reference's layout
    ({"model","optimizer","scheduler","iteration"}), `last_checkpoint`, and a second invocation that resumes from it --
    with the momentum buffers (a resume that silently restarted from zero momentum ends with visibly different buffers
    than the uninterrupted run)."""
    import shutil
    import subprocess
    opts = ["MODEL.WSOD_ON", "True", "MODEL.FASTER_RCNN", "False", "MODEL.BACKBONE.CONV_BODY", "VGG16-OICR",
            "MODEL.ROI_BOX_HEAD.NUM_CLASSES", "21", "MODEL.ROI_BOX_HEAD.POOLER_METHOD", "ROIPool",
            "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", "7", "MODEL.ROI_BOX_HEAD.POOLER_SCALES", "(0.125,)",
            "MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR", "VGG16.roi_head", "MODEL.ROI_WEAK_HEAD.PREDICTOR", "MISTPredictor",
            "MODEL.ROI_WEAK_HEAD.LOSS", "RoIRegLoss", "MODEL.ROI_WEAK_HEAD.REGRESS_ON", "True", "DB.METHOD", "dropblock",
            "SOLVER.CONTRA", "True", "SOLVER.BASE_LR", "1e-5", "SOLVER.CHECKPOINT_PERIOD", "2", "SEED", "7",
            "MODEL.WEIGHT", ""]
    env = dict(os.environ, ODW_NO_TIMER="1")

    def run(out_dir, max_iter):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "train_net.py"), "--synthetic", "--size", "160", "--proposals", "60",
               "--log-period", "1"] + opts + ["SOLVER.MAX_ITER", str(max_iter), "OUTPUT_DIR", out_dir]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return r.stdout

    straight, resumed = str(tmp_path / "a"), str(tmp_path / "b")
    run(straight, 4)
    for f in ("model_0000002.pth", "model_final.pth", "last_checkpoint"):
        assert os.path.exists(os.path.join(straight, f)), f
    ck2 = torch.load(os.path.join(straight, "model_0000002.pth"), weights_only=False)
    assert set(ck2) >= {"model", "optimizer", "scheduler", "iteration"} and ck2["iteration"] == 2
    assert len(ck2["optimizer"]["state"]) == len(ck2["optimizer"]["param_groups"]) > 40
    # second directory: starts from the iteration-2 checkpoint of the first run and continues to 4
    os.makedirs(resumed)
    shutil.copy(os.path.join(straight, "model_0000002.pth"), os.path.join(resumed, "model_0000002.pth"))
    with open(os.path.join(resumed, "last_checkpoint"), "w") as f:
        f.write(os.path.join(resumed, "model_0000002.pth"))
    out = run(resumed, 4)
    assert "resume, iteration 2" in out, out[-1500:]
    assert "iter: 3" in out and "iter: 4" in out and "iter: 2 " not in out
    a = torch.load(os.path.join(straight, "model_final.pth"), weights_only=False)
    b = torch.load(os.path.join(resumed, "model_final.pth"), weights_only=False)
    assert a["iteration"] == b["iteration"] == 4
    # Buffers that are rounding noise are left out: the detection stream's class bias sits under a softmax over the PROPOSALS, so
    # its true gradient is zero and its buffer holds ~1e-8 of summation-order noise (the other buffers: 1e-3 .. 1).  Its bits
    # follow the kernel plans, and a fresh process starts from the capacity-sized plans again (loss_device.HintReader) -- measured
    # (tools/exp/resume_diff.py): that one (21,) buffer differs by 0.48 of its 1.2e-8 norm, every other buffer by < 1e-2, and two
    # uninterrupted runs agree bit for bit.
    norms = {i: float(sa["momentum_buffer"].double().norm()) for i, sa in a["optimizer"]["state"].items()}
    floor = 1e-5 * max(norms.values())
    worst, compared = 0.0, 0
    for i, sa in a["optimizer"]["state"].items():
        ma, mb = sa["momentum_buffer"].double(), b["optimizer"]["state"][i]["momentum_buffer"].double()
        if norms[i] > floor:
            worst = max(worst, float((ma - mb).norm() / ma.norm()))
            compared += 1
    nonzero = sum(1 for v in norms.values() if v > 0)
    assert compared >= nonzero - 4 and compared > 20, (compared, nonzero, len(norms))
    assert worst < 0.05, worst           # (a zero-momentum restart is off by ~0.5; bf16 run-to-run noise is far below this)
