"""Parity of the gfx950 kernels (called through the C-ABI via od_wscl_amd._C) with the CPU
oracle and the reference-generated golden vectors.  Needs a real MI355X: `pytest -m gpu`.

Bars: bit-exact for index/byte work (ROIPool output + argmax, NMS keep lists, IoU selections,
ROIAlign vs the reference CPU kernel within 1 ulp-level tolerance stated below), fp32
tolerance for accumulations whose summation order is not defined by the reference (atomicAdd
backward passes, matrix products)."""
import numpy as np
import pytest
import torch

from od_wscl_amd import synthetic
from od_wscl_amd.utils import rng
from oracle import native

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def C():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import _C
    return _C


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rois_for(seed, B, n_per, H, W, min_size=8):
    parts = []
    for b in range(B):
        bx = synthetic.make_proposals(seed, b, n_per, H, W, min_size=min_size)
        parts.append(np.concatenate([np.full((n_per, 1), b, np.float32), bx], 1))
    return np.concatenate(parts, 0)


EDGE = np.array([[0, 0, 0, 95, 79], [0, 4, 4, 12, 12], [1, 3.9, 4.1, 60.2, 70.7], [1, 20, 20, 20, 20],
                 [0, 30, 30, 10, 10], [1, -40, -40, 20, 20], [0, 80, 60, 200, 200], [0, 1000, 1000, 1100, 1100],
                 [0, 2.5, 0.5, 2.5, 0.5], [1, 12, 4, 12.5, 4.5], [0, 11.5, 3.5, 44.5, 36.5]], np.float32)


@pytest.mark.parametrize("B,Cc,H,W,n,scale,ph,pw", [
    (2, 8, 20, 24, 64, 0.125, 7, 7),      # plane path, CG=1 (few channels)
    (1, 512, 38, 50, 300, 0.125, 7, 7),   # plane path, CG=2
    (2, 1024, 12, 16, 40, 0.0625, 7, 7),  # plane path, CG=4
    (1, 5, 10, 12, 30, 0.25, 3, 5),       # non-square pooling
    (1, 3, 210, 200, 50, 0.5, 7, 7),      # plane does not fit in LDS -> direct kernels
])
def test_roi_pool_forward_backward(C, B, Cc, H, W, n, scale, ph, pw):
    feat = rng.normal(21, B * Cc + H, B * Cc * H * W).reshape(B, Cc, H, W)
    rois = rois_for(22, B, n, int(H / scale), int(W / scale))
    rois = np.concatenate([rois, EDGE[EDGE[:, 0] < B]], 0)
    out, arg = C.roi_pool_forward(dev(feat), dev(rois), scale, ph, pw)
    ro, ra = native.roi_pool_fwd(feat, rois, scale, ph, pw)
    np.testing.assert_array_equal(arg.cpu().numpy(), ra)          # bit-exact index selection
    np.testing.assert_array_equal(out.cpu().numpy(), ro)
    g = rng.normal(23, 1, ro.size).reshape(ro.shape)
    gin = C.roi_pool_backward(dev(g), None, dev(rois), arg, scale, ph, pw, B, Cc, H, W)
    rg = native.roi_pool_bwd(g, ra, rois, feat.shape, ph, pw)
    # the reference sums with atomicAdd (order undefined): fp32 re-association tolerance
    np.testing.assert_allclose(gin.cpu().numpy(), rg, rtol=1e-5, atol=1e-5)
    # ours is the fixed-point (order-independent) form: bit-identical from run to run, and the exact sum to 2^-40 of
    # the largest gradient -- closer to the double-precision sum than any fp32 summation order
    gin2 = C.roi_pool_backward(dev(g), None, dev(rois), arg, scale, ph, pw, B, Cc, H, W)
    assert torch.equal(gin, gin2)
    # a heavy-collision case: every ROI is the full image, so each bin's arg-max cell receives n contributions
    full = np.tile(np.array([[0, 0, 0, W / scale - 1, H / scale - 1]], np.float32), (200, 1))
    o2, a2 = C.roi_pool_forward(dev(feat), dev(full), scale, ph, pw)
    g2 = rng.normal(24, 1, o2.numel()).reshape(tuple(o2.shape)) * np.exp(rng.normal(25, 1, o2.numel()).reshape(tuple(o2.shape)) * 4)
    runs = [C.roi_pool_backward(dev(g2), None, dev(full), a2, scale, ph, pw, B, Cc, H, W) for _ in range(3)]
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    want = np.zeros((B * Cc * H * W,), np.float64)
    idx = a2.cpu().numpy().reshape(200, Cc, -1)
    flat = g2.astype(np.float64).reshape(200, Cc, -1)
    for c in range(Cc):
        ok = idx[:, c] >= 0
        np.add.at(want, c * H * W + idx[:, c][ok], flat[:, c][ok])
    got = runs[0].cpu().numpy().reshape(-1).astype(np.float64)
    assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max() + 2.0 ** -38 * np.abs(g2).max()


def test_roi_pool_empty_and_known_answers(C):
    feat = np.arange(2 * 1 * 4 * 6, dtype=np.float32).reshape(2, 1, 4, 6)
    feat[1] = -feat[1]
    rois = np.array([[0, 0, 0, 5, 3], [1, 0, 0, 5, 3], [0, 2.5, 0.5, 2.5, 0.5], [0, 10, 10, 12, 12]], np.float32)
    out, arg = C.roi_pool_forward(dev(feat), dev(rois), 1.0, 2, 3)
    np.testing.assert_array_equal(arg[0, 0].cpu().numpy(), [[7, 9, 11], [19, 21, 23]])
    np.testing.assert_array_equal(arg[1, 0].cpu().numpy(), [[0, 2, 4], [12, 14, 16]])
    assert (arg[3] == -1).all() and (out[3] == 0).all()
    o0, a0 = C.roi_pool_forward(dev(feat), dev(rois[:0]), 1.0, 2, 3)
    assert o0.shape == (0, 1, 2, 3) and a0.shape == (0, 1, 2, 3)
    gin = C.roi_pool_backward(dev(np.zeros((0, 1, 2, 3), np.float32)), None, dev(rois[:0]), a0, 1.0, 2, 3, 2, 1, 4, 6)
    assert gin.shape == (2, 1, 4, 6) and (gin == 0).all()
    with pytest.raises(RuntimeError):
        C.roi_pool_forward(torch.from_numpy(feat), torch.from_numpy(rois), 1.0, 2, 3)   # CPU tensors: no fallback


def test_roi_align_forward_golden(C, ops_golden):
    g = ops_golden
    for sr in (0, 2):
        for scale in (0.125, 0.25):
            out = C.roi_align_forward(dev(g["ra_feat"]), dev(g["ra_rois"]), scale, 7, 7, sr).cpu().numpy()
            ref = g["ra_out_sr%d_s%g" % (sr, scale)]
            # same fp32 op sequence as ROIAlign_cpu.cpp (built -ffp-contract=off): expect equality; the
            # stated bar is 1e-6 abs on O(1) values in case the device's division differs by an ulp
            np.testing.assert_allclose(out, ref, rtol=0, atol=1e-6)
    out = C.roi_align_forward(dev(g["ra_feat"]), dev(g["ra_rois"]), 0.125, 3, 5, 0).cpu().numpy()
    np.testing.assert_allclose(out, g["ra_out_3x5"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("B,Cc,H,W,n,scale,sr", [
    (2, 8, 20, 24, 64, 0.125, 0), (1, 256, 38, 50, 120, 0.125, 2), (1, 3, 210, 200, 40, 0.5, 0)])
def test_roi_align_forward_backward_vs_oracle(C, B, Cc, H, W, n, scale, sr):
    feat = rng.normal(31, B * Cc + H, B * Cc * H * W).reshape(B, Cc, H, W)
    rois = rois_for(32, B, n, int(H / scale), int(W / scale))
    rois = np.concatenate([rois, EDGE[EDGE[:, 0] < B]], 0)
    out = C.roi_align_forward(dev(feat), dev(rois), scale, 7, 7, sr).cpu().numpy()
    ref = native.roi_align_fwd(feat, rois, scale, 7, 7, sr)
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-6)
    g = rng.normal(33, 1, ref.size).reshape(ref.shape)
    gin = C.roi_align_backward(dev(g), dev(rois), scale, 7, 7, B, Cc, H, W, sr).cpu().numpy()
    rg = native.roi_align_bwd(g, rois, scale, feat.shape, 7, 7, sr)
    np.testing.assert_allclose(gin, rg, rtol=1e-4, atol=1e-4)     # atomics order (ROIAlign_cuda.cu:246-249)


def test_nms_golden_and_modes(C, ops_golden):
    g = ops_golden
    b, s = dev(g["nms_boxes"]), dev(g["nms_scores"])
    for thr in (0.1, 0.3, 0.5, 0.7):
        np.testing.assert_array_equal(C.nms_cpu_rule(b, s, thr).cpu().numpy(), g["nms_keep_ge_%g" % thr])
        np.testing.assert_array_equal(C.nms(b, s, thr).cpu().numpy(), native.nms_wt(g["nms_boxes"], g["nms_scores"], thr, False))
        np.testing.assert_array_equal(C.nms_torchvision(b, s, thr).cpu().numpy(), native.nms_tv(g["nms_boxes"], g["nms_scores"], thr))
    eb, es = dev(g["nms_eq_boxes"]), dev(g["nms_eq_scores"])
    assert C.nms_cpu_rule(eb, es, 0.5).tolist() == [0, 2] and C.nms(eb, es, 0.5).tolist() == [0, 1, 2]
    tb = dev(np.array([[0, 0, 10, 10], [0, 0, 10, 5], [0, 0, 10, 10], [20, 20, 30, 30]], np.float32))
    ts = dev(np.array([0.5, 0.9, 0.5, 0.1], np.float32))
    assert C.nms_torchvision(tb, ts, 0.5).tolist() == [1, 0, 3] and C.nms_torchvision(tb, ts, 0.49).tolist() == [1, 3]
    assert C.nms_torchvision(tb[:0], ts[:0], 0.5).numel() == 0
    assert C.nms_torchvision(tb[:1], ts[:1], 0.5).tolist() == [0]


@pytest.mark.parametrize("n", [63, 64, 65, 1000, 4000, 8192])
def test_nms_sizes(C, n):
    boxes = synthetic.make_proposals(40 + n, 0, n, 1000, 1500, min_size=6)
    scores = rng.uniform(41, n, n)
    scores[: n // 8] = np.round(scores[: n // 8] * 8) / 8         # many exact ties
    for thr in (0.1, 0.5):
        np.testing.assert_array_equal(C.nms_torchvision(dev(boxes), dev(scores), thr).cpu().numpy(),
                                      native.nms_tv(boxes, scores, thr))
    np.testing.assert_array_equal(C.nms_cpu_rule(dev(boxes), dev(scores), 0.3).cpu().numpy(),
                                  native.nms_wt(boxes, scores, 0.3, True))


def test_box_iou(C, ops_golden):
    g = ops_golden
    np.testing.assert_array_equal(C.box_iou(dev(g["iou_a"]), dev(g["iou_b"])).cpu().numpy(), g["iou_ab"])
    a = synthetic.make_proposals(50, 0, 2000, 600, 600)
    np.testing.assert_array_equal(C.box_iou(dev(a), dev(a[:3])).cpu().numpy(), native.box_iou(a, a[:3]))
    sel = (C.box_iou(dev(g["iou_a"]), dev(g["iou_a"][3:4])) >= 0.5).nonzero()[:, 0].cpu().numpy()
    np.testing.assert_array_equal(sel, g["cal_iou_idx"])


@pytest.mark.parametrize("P", [1, 31, 64, 100, 500, 2000])
def test_pairwise_sim(C, P):
    E = rng.normal(60, P, P * 128).reshape(P, 128)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    S = C.pairwise_sim(dev(E)).cpu().numpy()
    ref = E.astype(np.float64) @ E.astype(np.float64).T
    assert np.abs(S - ref).max() <= 2e-6                           # exact-fp32 MFMA chain, |S| <= 1
    assert np.array_equal(S, S.T)                                  # mirrored tiles are the same numbers
    if P <= 500:
        np.testing.assert_allclose(S, native.pairwise_sim(E), rtol=0, atol=2e-6)
    E48 = rng.normal(61, P, P * 48).reshape(P, 48)                 # generic-D path
    np.testing.assert_allclose(C.pairwise_sim(dev(E48)).cpu().numpy(), E48 @ E48.T, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("P", [4000, 4001, 8000])
def test_pairwise_sim_at_the_benchmarked_sizes(C, P):
    """The sizes bench.py measures and claims a roofline fraction at (P = 4000: the COCO proposal count, 8000; 4001: a
    ragged last panel), against float64 (weak_head/loss.py:319: sim_mat = E E^T)."""
    E = rng.normal(62, P, P * 128).reshape(P, 128)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    S = C.pairwise_sim(dev(E)).cpu().numpy()
    assert S.shape == (P, P)
    E64 = E.astype(np.float64)
    worst = 0.0
    for r0 in range(0, P, 1000):                                          # fp64 reference in row panels
        worst = max(worst, float(np.abs(S[r0:r0 + 1000] - E64[r0:r0 + 1000] @ E64.T).max()))
    assert worst <= 2e-6, worst                                           # exact-fp32 MFMA chain, |S| <= 1
    assert np.array_equal(S, S.T)
    assert np.abs(np.diag(S) - 1.0).max() <= 1e-6


@pytest.mark.parametrize("P", [1, 31, 33, 225, 1000, 3001, 5000])
def test_pairwise_padded_pitch_is_bit_identical_and_leaves_the_padding_alone(C, P):
    """odw_pairwise_sim_ld (rows of S padded to 32 floats: C.pairwise_sim(padded=True) returns a view of it) against the dense entry:
    the same bits in the (P, P) view, and the padding columns are never written."""
    from od_wscl_amd import _lib as L
    lib = L.lib()
    E = rng.normal(64, P, P * 128).reshape(P, 128)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    Et = dev(E)
    ld = -(-P // 32) * 32
    S0 = torch.full((P, P), float("nan"), device="cuda")
    S1 = torch.full((P, ld), float("nan"), device="cuda")
    L.check(lib.odw_pairwise_sim(L.ptr(Et), P, 128, L.ptr(S0), L.stream()), "dense")
    L.check(lib.odw_pairwise_sim_ld(L.ptr(Et), P, 128, L.ptr(S1), ld, L.stream()), "padded")
    assert not torch.isnan(S0).any()
    assert torch.equal(S0, S1[:, :P])
    assert torch.isnan(S1[:, P:]).all()
    V = C.pairwise_sim(Et, padded=True)
    assert V.shape == (P, P) and V.stride(0) == ld and torch.equal(V, S0)
    D = C.pairwise_sim(Et)
    assert D.is_contiguous() and torch.equal(D, S0)
    with pytest.raises(RuntimeError):
        L.check(lib.odw_pairwise_sim_ld(L.ptr(Et), P, 128, L.ptr(S1), P - 1, L.stream()), "short pitch")


@pytest.mark.parametrize("P", [1, 31, 33, 224, 225, 1000, 2000, 4001])
def test_pairwise_planes_form_is_bit_identical_to_the_one_launch_form(P):
    """odw_pairwise_sim_ws with a workspace (split kernel + planes by LDS-DMA) and the caller-planes entry against the
    one-launch panel kernel: the same six plane products in the same order -> the same bits; the planes themselves are
    the exact decomposition hi + mid + lo == x with zero padding rows."""
    from od_wscl_amd import _lib as L
    lib = L.lib()
    E = rng.normal(63, P, P * 128).reshape(P, 128)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    Et = dev(E)
    S0, S1, S2 = (torch.full((P, P), float("nan"), device="cuda") for _ in range(3))
    L.check(lib.odw_pairwise_sim(L.ptr(Et), P, 128, L.ptr(S0), L.stream()), "panel")
    wsb = lib.odw_pairwise_sim_workspace(P, 128)
    ws = torch.full((wsb,), 0xFF, dtype=torch.uint8, device="cuda")
    L.check(lib.odw_pairwise_sim_ws(L.ptr(Et), P, 128, L.ptr(S1), L.ptr(ws), wsb, L.stream()), "ws")
    assert torch.equal(S0, S1) and not torch.isnan(S1).any()
    ws2 = torch.full((wsb,), 0xFF, dtype=torch.uint8, device="cuda")
    L.check(lib.odw_pairwise_split_planes(L.ptr(Et), P, L.ptr(ws2), L.stream()), "split")
    L.check(lib.odw_pairwise_sim_planes(L.ptr(ws2), P, L.ptr(S2), L.stream()), "planes")
    assert torch.equal(S0, S2)
    ppad = (P + 31) // 32 * 32
    pl = ws2[:3 * ppad * 256].view(torch.bfloat16).view(3, ppad, 128).float().cpu().numpy()
    np.testing.assert_array_equal(pl[0, :P] + pl[1, :P] + pl[2, :P], E)           # exact (each partial sum is representable)
    assert not pl[:, P:].any()


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_supcon_golden(C, ops_golden, name):
    g = ops_golden
    F_, y, w = g["sc_%s_F" % name], g["sc_%s_labels" % name], g["sc_%s_w" % name]
    loss, dF = C.supcon_v2(dev(F_), dev(y), dev(w), 0.2)
    ref = float(g["sc_%s_loss" % name])
    assert abs(loss.item() - ref) <= 1e-5 * abs(ref)              # well inside north_star's 1e-3 rel
    rd = g["sc_%s_dF" % name]
    assert np.abs(dF.cpu().numpy() - rd).max() <= 1e-4 * np.abs(rd).max()


@pytest.mark.parametrize("N,ncls", [(1, 1), (2, 1), (33, 4), (185, 2), (1000, 20), (4000, 80)])
def test_supcon_vs_oracle(C, N, ncls):
    F_ = rng.normal(70, N, N * 128).reshape(N, 128)
    F_ /= np.linalg.norm(F_, axis=1, keepdims=True)
    y = (rng.uniform(71, N, N) * ncls).astype(np.int32)
    w = rng.uniform(72, N, N)
    loss, dF = C.supcon_v2(dev(F_), dev(y), dev(w), 0.2, grad_scale=0.03)
    rl, rd = native.supcon_v2(F_, y, w, 0.2)
    if np.isfinite(rl):
        assert abs(loss.item() - rl) <= 2e-5 * abs(rl)
        assert np.abs(dF.cpu().numpy() - 0.03 * rd).max() <= 2e-4 * max(np.abs(0.03 * rd).max(), 1e-12)
    else:       # a label that occurs once gives -log 0 = inf, like torch
        assert not np.isfinite(loss.item())


def test_layers_autograd(C):
    from od_wscl_amd.layers import ROIPool, ROIAlign
    feat = rng.normal(80, 1, 1 * 16 * 20 * 24).reshape(1, 16, 20, 24)
    rois = rois_for(81, 1, 30, 160, 192)
    for layer, fwd, bwd in ((ROIPool((7, 7), 0.125), None, None), (ROIAlign((7, 7), 0.125, 0), None, None)):
        x = dev(feat).requires_grad_(True)
        out = layer(x, dev(rois))
        g = rng.normal(82, 1, out.numel()).reshape(tuple(out.shape))
        out.backward(dev(g))
        if isinstance(layer, ROIPool):
            ro, ra = native.roi_pool_fwd(feat, rois, 0.125, 7, 7)
            rg = native.roi_pool_bwd(g, ra, rois, feat.shape, 7, 7)
        else:
            ro = native.roi_align_fwd(feat, rois, 0.125, 7, 7, 0)
            rg = native.roi_align_bwd(g, rois, 0.125, feat.shape, 7, 7, 0)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ro, rtol=0, atol=2e-6)
        np.testing.assert_allclose(x.grad.cpu().numpy(), rg, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_roi_backward_propagates_non_finite_gradients(bad):
    """A NaN or inf in grad_out must reach grad_in as NaN (what a float sum would give), not be filed under "all-zero
    gradient" by the fixed-point scale (csrc/odw_fixed.h: the absmax pre-pass sorts NaN above inf)."""
    from od_wscl_amd import _C
    from od_wscl_amd import synthetic
    torch.manual_seed(0)
    feat = torch.randn(1, 8, 20, 24, device="cuda")
    boxes = synthetic.make_proposals(5, 0, 32, 160, 192, min_size=12)
    rois = torch.from_numpy(np.concatenate([np.zeros((32, 1), np.float32), boxes], 1)).cuda()
    out, arg = _C.roi_pool_forward(feat, rois, 0.125, 7, 7)
    g = torch.randn_like(out)
    g[3, 2, 1, 1] = bad
    gin = _C.roi_pool_backward(g, feat, rois, arg, 0.125, 7, 7, 1, 8, 20, 24)
    assert torch.isnan(gin).any(), "ROIPool backward swallowed a non-finite gradient"
    ga = _C.roi_align_backward(g, rois, 0.125, 7, 7, 1, 8, 20, 24, 2)
    assert torch.isnan(ga).any(), "ROIAlign backward swallowed a non-finite gradient"
    gin0 = _C.roi_pool_backward(torch.zeros_like(out), feat, rois, arg, 0.125, 7, 7, 1, 8, 20, 24)
    assert float(gin0.abs().sum()) == 0.0
