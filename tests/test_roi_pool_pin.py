"""The ROIPool oracle pinned from a second side.

The reference has no CPU ROIPool (wetectron/csrc/ROIPool.h:23), so the C oracle's pooling is what the imported reference
runs inside every end-to-end golden.  oracle/roi_pool_numpy.py restates wetectron/csrc/cuda/ROIPool_cuda.cu:17-108 a
second time, in plain numpy and independently of odw_oracle.c; here it is swept over > 10^4 random and adversarial
ROIs against (CPU) the C oracle and (GPU) the HIP operator: output values and first-maximum positions bit for bit.

Adversarial families: corners that land on x.5 after scaling (C round() is half away from zero, numpy / torch round half
to even), negative and far-outside coordinates, zero-extent and inverted boxes, sub-cell boxes, boxes thinner than the
bin grid (empty bins), and feature maps with few distinct values (ties everywhere: the strict '>' scan must keep the
FIRST maximum), +0.0 / -0.0 windows and windows that hold nothing above -FLT_MAX."""
import numpy as np
import pytest

from oracle import native
from oracle import roi_pool_numpy as RP


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def make_rois(seed, n, B, H, W, scale):
    """n ROIs (n, 5) over a (H, W) map at `scale`: 40 % plain random boxes, 60 % adversarial families."""
    g = _rng(seed)
    iw, ih = W / scale, H / scale               # image extent in ROI coordinates
    step = 1.0 / scale
    out = np.zeros((n, 5), np.float32)
    out[:, 0] = g.integers(0, B, n)
    fam = g.integers(0, 10, n)
    for k in range(n):
        f = fam[k]
        if f < 4:                                # plain: integer corners anywhere inside (the data pipeline's boxes)
            x1, y1 = g.integers(0, int(iw) - 2), g.integers(0, int(ih) - 2)
            x2, y2 = g.integers(x1, int(iw)), g.integers(y1, int(ih))
        elif f == 4:                             # every corner lands on k + 0.5 after scaling (positive AND negative k)
            x1, y1 = (g.integers(-3, W - 1) + 0.5) * step, (g.integers(-3, H - 1) + 0.5) * step
            x2, y2 = (g.integers(-3, W + 3) + 0.5) * step, (g.integers(-3, H + 3) + 0.5) * step
        elif f == 5:                             # partly or wholly outside, negative coordinates
            x1, y1 = g.uniform(-2 * iw, iw), g.uniform(-2 * ih, ih)
            x2, y2 = x1 + g.uniform(0, 3 * iw), y1 + g.uniform(0, 3 * ih)
        elif f == 6:                             # zero extent / inverted
            x1, y1 = g.uniform(0, iw), g.uniform(0, ih)
            x2, y2 = x1 - g.integers(0, 3) * g.uniform(0, 40), y1 - g.integers(0, 3) * g.uniform(0, 40)
        elif f == 7:                             # smaller than one cell, fractional corners
            x1, y1 = g.uniform(0, iw), g.uniform(0, ih)
            x2, y2 = x1 + g.uniform(0, step), y1 + g.uniform(0, step)
        elif f == 8:                             # a few cells wide / tall: fewer cells than bins on one or both axes
            x1, y1 = g.uniform(0, iw - 1), g.uniform(0, ih - 1)
            x2, y2 = x1 + g.uniform(0, 6) * step, y1 + g.uniform(0, 40) * step
        else:                                    # corners within 1e-3 of a rounding boundary
            x1, y1 = (g.integers(0, W) + 0.5) * step + g.choice([-1e-3, 0, 1e-3]), (g.integers(0, H) + 0.5) * step
            x2, y2 = (g.integers(0, W) + 0.5) * step, (g.integers(0, H) + 0.5) * step + g.choice([-1e-3, 0, 1e-3])
        out[k, 1:] = (x1, y1, x2, y2)
    return out


def make_feat(seed, B, C, H, W, kind):
    g = _rng(seed)
    if kind == "normal":
        return g.standard_normal((B, C, H, W)).astype(np.float32)
    if kind == "ties":                           # four distinct values: nearly every window has a tied maximum
        return g.integers(-2, 2, (B, C, H, W)).astype(np.float32)
    if kind == "zeros":                          # +0.0 / -0.0 only: '>' treats them as equal, the first cell wins
        f = np.zeros((B, C, H, W), np.float32)
        f[g.random((B, C, H, W)) < 0.5] = np.float32(-0.0)
        return f
    f = g.standard_normal((B, C, H, W)).astype(np.float32)          # "lowest": patches of -FLT_MAX (never beat the start value)
    f[g.random((B, C, H, W)) < 0.6] = RP._F32_LOWEST
    return f


SWEEP = [  # (seed, ROIs, B, C, H, W, scale, ph, pw, feature kind)
    (1, 4000, 2, 8, 40, 56, 0.125, 7, 7, "normal"),
    (2, 4000, 2, 8, 40, 56, 0.125, 7, 7, "ties"),
    (3, 1500, 1, 8, 24, 32, 0.0625, 7, 7, "zeros"),
    (4, 1500, 2, 8, 24, 32, 0.25, 7, 7, "lowest"),
    (5, 600, 1, 3, 20, 28, 0.125, 3, 5, "ties"),          # non-square pooling, C % 8 != 0 (the plane kernels on the GPU)
]


def test_sweep_is_large_and_adversarial():
    assert sum(s[1] for s in SWEEP) >= 10000
    r = make_rois(1, 4000, 2, 40, 56, 0.125)
    sx = r[:, 1] * np.float32(0.125)
    assert (np.abs(sx - np.floor(sx) - 0.5) < 1e-6).sum() > 300          # x.5 after scaling
    assert (r[:, 1] < 0).sum() > 100 and (r[:, 3] < r[:, 1]).sum() > 100 and (r[:, 3] == r[:, 1]).sum() > 20


def test_half_away_from_zero_rounding_not_half_even():
    x = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 2.4999998, -2.4999998], np.float32)
    np.testing.assert_array_equal(RP._round_half_away(x), [1, 2, 3, -1, -2, -3, 2, -2])


@pytest.mark.parametrize("case", SWEEP, ids=lambda c: "seed%d-%s" % (c[0], c[-1]))
def test_numpy_restatement_equals_the_c_oracle(case):
    seed, n, B, C, H, W, scale, ph, pw, kind = case
    feat = make_feat(seed, B, C, H, W, kind)
    rois = make_rois(100 + seed, n, B, H, W, scale)
    out_np, arg_np = RP.roi_pool_forward(feat, rois, scale, ph, pw)
    out_c, arg_c = native.roi_pool_fwd(feat, rois, scale, ph, pw)
    np.testing.assert_array_equal(arg_np, arg_c)
    np.testing.assert_array_equal(out_np.view(np.uint32), out_c.view(np.uint32))       # bit for bit, the sign of zero included
    assert (arg_np == -1).any() and (arg_np >= 0).any()
    # backward: the C oracle's fp32 scatter against the exact (float64) sums
    g = _rng(200 + seed).standard_normal(out_np.shape).astype(np.float32)
    gin_c = native.roi_pool_bwd(g, arg_c, rois, feat.shape, ph, pw)
    gin_np = RP.roi_pool_backward(g, arg_np, rois, feat.shape)
    scale_ = max(1.0, float(np.abs(gin_np).max()))
    assert np.abs(gin_c - gin_np).max() <= 2e-6 * scale_ * max(1.0, np.sqrt(n / 100.0))


@pytest.mark.gpu
@pytest.mark.parametrize("case", SWEEP, ids=lambda c: "seed%d-%s" % (c[0], c[-1]))
def test_hip_operator_equals_the_numpy_restatement(case):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import _C
    seed, n, B, C, H, W, scale, ph, pw, kind = case
    feat = make_feat(seed, B, C, H, W, kind)
    rois = make_rois(100 + seed, n, B, H, W, scale)
    out_np, arg_np = RP.roi_pool_forward(feat, rois, scale, ph, pw)
    out, arg = _C.roi_pool_forward(torch.from_numpy(feat).cuda(), torch.from_numpy(rois).cuda(), scale, ph, pw)
    np.testing.assert_array_equal(arg.cpu().numpy(), arg_np)
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got, out_np)                   # every value equal ...
    # ... and bit for bit, with ONE documented exception: the kernels pool on an order-preserving integer image of the
    # map in which -0.0 is folded onto +0.0 (the reference's '>' treats them as equal, so the first cell wins either
    # way: the POSITIONS above are exact); where the winning cell holds -0.0 the kernel returns +0.0
    diff = got.view(np.uint32) != out_np.view(np.uint32)
    assert not diff.any() or ((out_np[diff] == 0).all() and (got.view(np.uint32)[diff] == 0).all()
                              and (out_np.view(np.uint32)[diff] == 0x80000000).all())
    if kind != "zeros":
        assert not diff.any()
    g = _rng(200 + seed).standard_normal(out_np.shape).astype(np.float32)
    gin = _C.roi_pool_backward(torch.from_numpy(g).cuda(), None, torch.from_numpy(rois).cuda(), arg, scale, ph, pw, B, C, H, W)
    gin_np = RP.roi_pool_backward(g, arg_np, rois, feat.shape)
    # the HIP backward sums in 64-bit fixed point (exact to 2^-40 of the largest term): closer to the exact sum than
    # one fp32 rounding of the result
    scale_ = max(1.0, float(np.abs(gin_np).max()))
    assert np.abs(gin.cpu().numpy() - gin_np).max() <= 2e-7 * scale_
