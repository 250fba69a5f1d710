"""csrc/split.hip + the split precision modes (od_wscl_amd/precision.py): fp32-grade Linear / convolution products on
the bf16 matrix cores.  References: numpy for the plane decomposition (bit-exact), torch fp64 for the products --
the bar is what an fp32 GEMM itself achieves (a few 2^-24 * sqrt(K) of the accumulated magnitude)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf16_rn(x):
    """numpy round-to-nearest-even fp32 -> bf16 (as fp32 values)."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _planes(x):
    hi = _bf16_rn(x)
    r1 = (x - hi).astype(np.float32)
    mid = _bf16_rn(r1)
    lo = _bf16_rn((r1 - mid).astype(np.float32))
    return [hi, mid, lo, np.zeros_like(hi)]


@pytest.mark.parametrize("R,C,block", [(5, 24, 64), (130, 100, 128), (64, 4096, 4096), (33, 7, 8)])
def test_split_rows_and_cols_are_the_plane_decomposition(R, C, block):
    from od_wscl_amd import precision as P
    rs = np.random.RandomState(R * 1000 + C)
    x = (rs.randn(R, C) * np.exp(rs.randn(R, C) * 3)).astype(np.float32)
    x[0, 0] = 0.0
    pl = _planes(x)
    assert np.all(pl[0].astype(np.float64) + pl[1] + pl[2] == x.astype(np.float64))       # three bf16 planes hold an fp32 exactly
    xt = torch.from_numpy(x).cuda()
    for pat in [(0, 0, 0, 1, 1, 2), (0, 1, 2, 0, 1, 0), (2,), (0, 0, 0, 1, 1, 2, 3, 3)]:
        out = P.split_rows(xt, pat, block).float().cpu().numpy()
        assert out.shape == (R, len(pat) * block)
        for t, p in enumerate(pat):
            np.testing.assert_array_equal(out[:, t * block:t * block + C], pl[p])
            assert not out[:, t * block + C:(t + 1) * block].any()
        rb = (R + 63) // 64 * 64
        outc = P.split_cols(xt, pat, rb).float().cpu().numpy()
        assert outc.shape == (C, len(pat) * rb)
        for t, p in enumerate(pat):
            np.testing.assert_array_equal(outc[:, t * rb:t * rb + R], pl[p].T)
            assert not outc[:, t * rb + R:(t + 1) * rb].any()


@pytest.mark.parametrize("mode,tol", [("bf16x3", 1e-6), ("bf16x2", 3e-5)])
@pytest.mark.parametrize("M,N,K", [(300, 357, 4096), (96, 4096, 25088), (1000, 128, 4096)])
def test_split_linear_matches_fp64(M, N, K, mode, tol):
    """forward, input gradient, weight gradient and bias gradient of the fused Linear against torch fp64."""
    from od_wscl_amd import gemm, precision as P
    P.set_precision(mode)
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).requires_grad_(True)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).requires_grad_(True)
    b = torch.randn(N, device="cuda", generator=g).requires_grad_(True)
    gy = torch.randn(M, N, device="cuda", generator=g)
    y = gemm.fused_linear(x, w, b, gemm.Shadow(w), relu=True)
    assert y.dtype == torch.float32
    y.backward(gy)
    xd, wd, bd = x.detach().double(), w.detach().double(), b.detach().double()
    pre = xd @ wd.t() + bd
    yd = pre.clamp_min(0)
    scale = float(np.sqrt(K)) * x.detach().abs().mean().item() * w.detach().abs().mean().item()     # accumulated magnitude
    if mode == "bf16x3":        # fp32 accumulation noise of a K' = 6K long MFMA chain (16 products per step), as any fp32 GEMM has
        tol = max(tol, 1.2e-7 * float(np.sqrt(6 * K / 16)))
    assert (y.double() - yd).abs().max().item() <= tol * scale * 8
    # (entries within rounding noise of the ReLU kink may land on either side: compare where |pre| is clear of it)
    clear = (pre.abs() > 1e-4 * scale).double()
    dz = gy.double() * (pre > 0).double()
    got_mask = (y != 0).double()
    assert ((got_mask - (pre > 0).double()).abs() * clear).sum().item() == 0
    dzg = gy.double() * got_mask                  # same mask as the kernel used
    ex, ew, eb = dzg @ wd, dzg.t() @ xd, dzg.sum(0)
    for got, exp, kk in ((x.grad, ex, N), (w.grad, ew, M), (b.grad, eb, M)):
        s = float(np.sqrt(kk)) * exp.abs().mean().item() + 1e-30
        assert (got.double() - exp).abs().max().item() <= 200 * tol * max(s, exp.abs().max().item()), (kk, s)
    del dz


@pytest.mark.parametrize("cin,cout,dil,H,W", [(3, 64, 1, 40, 48), (64, 128, 1, 24, 20), (256, 256, 2, 20, 24)])
def test_split_conv_matches_fp64(cin, cout, dil, H, W):
    """The VGG body in bf16x3 (one conv + ReLU layer at a time) against torch fp64 conv2d, forward and backward."""
    from od_wscl_amd import precision as P
    from od_wscl_amd.modeling.backbone.vgg16_hip import VGGBackboneHip
    P.set_precision("bf16x3")

    class Body(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.features = torch.nn.Sequential(
                torch.nn.Conv2d(cin, cout, 3, padding=dil, dilation=dil), torch.nn.ReLU(),
                torch.nn.Conv2d(cout, cout, 3, padding=1), torch.nn.ReLU(), torch.nn.MaxPool2d(2, 2),
                torch.nn.Conv2d(cout, 64, 3, padding=1))

    torch.manual_seed(cin)
    body = Body().cuda()
    hip = VGGBackboneHip(body)
    x = torch.randn(2, cin, H, W, device="cuda") * 30
    feat = hip(x)[0]
    gout = torch.randn_like(feat)
    feat.backward(gout)
    got = {n: p.grad.clone() for n, p in body.named_parameters()}
    for p in body.parameters():
        p.grad = None
    bd = Body().cuda().double()
    bd.load_state_dict({k: v.double() for k, v in body.state_dict().items()})
    from od_wscl_amd.layers.misc import library_reference
    with library_reference():                      # torch's float64 convolutions: the reference, not the product
        ref = bd.features(x.double())
    ref.backward(gout.double())
    err = (feat.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-6, err
    for n, p in bd.named_parameters():
        if n.startswith("features.0."):
            continue           # the first convolution receives no gradient request below it, but its own dW is computed
        e = (got[n].double() - p.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-30)
        assert e <= 2e-5, (n, e)
    e0 = (got["features.0.weight"].double() - bd.features[0].weight.grad).abs().max().item() / bd.features[0].weight.grad.abs().max().item()
    assert e0 <= 2e-5, e0


def test_bwd_mask_kernel_bias_gradient_is_deterministic():
    from od_wscl_amd import precision as P
    g = torch.Generator(device="cuda").manual_seed(3)
    dy = torch.randn(777, 130, device="cuda", generator=g)
    y = torch.randn(777, 130, device="cuda", generator=g).clamp_min(0)
    outs = []
    for _ in range(3):
        db = torch.zeros(130, device="cuda")
        dz = P.bwd_mask(dy, y, 2.0, db)
        outs.append((dz.clone(), db.clone()))
    exp = dy * (y != 0) * 2.0
    assert torch.equal(outs[0][0], exp)
    torch.testing.assert_close(outs[0][1], exp.sum(0), rtol=1e-5, atol=1e-4)
    assert all(torch.equal(outs[0][1], o[1]) for o in outs[1:])
