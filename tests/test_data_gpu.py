"""The pixel half of the data boundary on the MI355X: csrc/preprocess.hip through the C-ABI against the oracle
(Pillow + numpy) -- bit-exact -- and the whole dataset -> transforms -> BatchCollator -> .to(cuda) chain against the
batch the imported reference produced (tests/golden/data_voc.npz)."""
import os
import random

import numpy as np
import pytest
import torch

import voc_fixture
from oracle import data_ref as D
from test_data_cpu import _devkit
from yacs_like import cfg_for_data

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
MEAN, STD = [102.9801, 115.9465, 122.7717], [1.0, 1.0, 1.0]


def _run(pixels, out_hw, pad_hw, **kw):
    from od_wscl_amd import _C
    dev = torch.device("cuda:0")
    out = torch.full((3,) + tuple(pad_hw), 7.0, dtype=torch.float32, device=dev)     # stale content must be overwritten
    _C.image_preprocess(torch.from_numpy(pixels).to(dev), out_hw, out, kw.pop("mean", MEAN), kw.pop("std", STD), **kw)
    return out.cpu().numpy()


def _expect(pixels, out_hw, pad_hw, mean=MEAN, std=STD, to_bgr255=True, hflip=False, vflip=False, lighting=None):
    x = D.pixel_chain(pixels, out_hw, hflip, vflip, lighting, mean, std, to_bgr255)
    full = np.zeros((3,) + tuple(pad_hw), np.float32)
    full[:, : out_hw[0], : out_hw[1]] = x
    return full


def test_resize_is_pillow_bit_for_bit_over_shapes():
    rng = np.random.default_rng(3)
    cases = [((60, 80), (64, 85)), ((75, 50), (96, 64)), ((64, 64), (64, 64)), ((64, 64), (48, 48)),
             ((37, 91), (37, 200)), ((37, 91), (111, 91)), ((200, 150), (31, 23)), ((9, 7), (300, 233)),
             ((375, 500), (600, 800)), ((500, 333), (1200, 799)), ((480, 640), (176, 234))]
    for (h, w), (oh, ow) in cases:
        pixels = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        pad = (oh + 5, ow + 9)
        np.testing.assert_array_equal(_run(pixels, (oh, ow), pad), _expect(pixels, (oh, ow), pad), err_msg=str(((h, w), (oh, ow))))


def test_flips_lighting_and_normalisation_variants():
    rng = np.random.default_rng(4)
    pixels = rng.integers(0, 256, (53, 71, 3), dtype=np.uint8)
    light = np.array([0.013, -0.021, 0.007], np.float32)
    for kw in (dict(hflip=True), dict(vflip=True), dict(hflip=True, vflip=True), dict(lighting=light),
               dict(lighting=light, hflip=True), dict(to_bgr255=False, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]),
               dict(std=[57.375, 57.12, 58.395])):
        np.testing.assert_array_equal(_run(pixels, (80, 107), (96, 128), **dict(kw)), _expect(pixels, (80, 107), (96, 128), **kw),
                                      err_msg=str(kw))


def test_bad_arguments_fail_loudly():
    from od_wscl_amd import _C
    dev = torch.device("cuda:0")
    px = torch.zeros((8, 8, 3), dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError):       # padded plane smaller than the image
        _C.image_preprocess(px, (16, 16), torch.empty((3, 8, 8), device=dev), MEAN, STD)
    with pytest.raises(ValueError):
        _C.image_preprocess(px.float(), (16, 16), torch.empty((3, 16, 16), device=dev), MEAN, STD)
    with pytest.raises(RuntimeError):       # no CPU path
        _C.image_preprocess(px.cpu(), (16, 16), torch.empty((3, 16, 16)), MEAN, STD)


@pytest.mark.parametrize("mode", ["train", "test"])
def test_collated_batch_equals_the_reference_batch(tmp_path, mode):
    golden = np.load(os.path.join(HERE, "golden", "data_voc.npz"))
    root, pkl, images, proposals, ids = _devkit(tmp_path, golden)
    from od_wscl_amd.data import BatchCollator, build_transforms
    from od_wscl_amd.data.datasets import PascalVOCDataset
    is_train = mode == "train"
    ds = PascalVOCDataset(root, "trainval", use_difficult=not is_train,
                          transforms=build_transforms(cfg_for_data(golden), is_train), proposal_file=pkl)
    random.seed(int(golden["spec_seed"]))
    torch.manual_seed(int(golden["spec_seed"]))
    pending, targets, rois, idx = BatchCollator(32)([ds[i] for i in range(len(ids))])
    batch = pending.to("cuda:0")
    assert batch.tensors.is_cuda and [tuple(s) for s in batch.image_sizes] == [tuple(s) for s in golden[mode + "_image_sizes"].tolist()]
    np.testing.assert_array_equal(batch.tensors.cpu().numpy(), golden[mode + "_batch"])
    # a second materialisation (staging buffers reused) gives the same batch
    np.testing.assert_array_equal(pending.to("cuda:0").tensors.cpu().numpy(), golden[mode + "_batch"])
