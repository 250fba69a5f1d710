"""Implicit-GEMM convolutions and the NHWC bf16 VGG16 backbone vs PyTorch references.  -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from od_wscl_amd.utils import rng

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _bf16_precision():
    """These tests pin the bf16 kernels against fp32 references on bf16-rounded operands (the fp32-grade split mode
    has its own file, test_split_gpu.py)."""
    from od_wscl_amd import precision
    precision.set_precision("bf16")
    yield


def rnd(seed, shape, scale=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy((rng.normal(seed, 1, n) * scale).reshape(shape)).cuda()


def r64(n):
    return (n + 63) // 64 * 64


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import _lib
    return _lib


@pytest.mark.parametrize("B,Cin,Cout,H,W,dil,relu", [(1, 3, 64, 20, 28, 1, True), (2, 64, 64, 16, 24, 1, True),
                                                    (1, 128, 256, 19, 13, 2, False), (1, 512, 512, 12, 10, 2, True)])
def test_conv3x3_forward_and_input_gradient(lib, B, Cin, Cout, H, W, dil, relu):
    L = lib
    cp = max(8, 1 << (Cin - 1).bit_length())
    x = rnd(1, (B, Cin, H, W)).bfloat16().float()
    w = rnd(2, (Cout, Cin, 3, 3), 0.05).bfloat16().float()
    b = rnd(3, (Cout,), 0.1)
    zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    xn = torch.empty((B * H * W, cp), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_nchw_f32_to_nhwc_bf16(L.ptr(x), B, H * W, Cin, cp, L.ptr(xn), L.stream()), "to nhwc")
    wk = torch.empty((Cout, r64(9 * cp)), dtype=torch.bfloat16, device="cuda")
    wd = torch.empty((Cin, r64(9 * Cout)), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_conv_weight_prep(L.ptr(w), Cout, Cin, cp, L.ptr(wk), wk.stride(0), L.ptr(wd), wd.stride(0), L.stream()), "prep")
    y = torch.empty((B * H * W, Cout), dtype=torch.float32, device="cuda")
    L.check(L.lib().odw_conv3x3_nhwc_bf16(L.ptr(xn), B * H * W, H, W, cp, dil, 0, L.ptr(wk), wk.stride(0), Cout, L.ptr(y), Cout, 0,
                                          L.ptr(b), 1 if relu else 0, None, 0, L.ptr(zero), L.stream()), "conv")
    ref = F.conv2d(x, w, b, padding=dil, dilation=dil)
    if relu:
        ref = torch.relu(ref)
    got = y.reshape(B, H, W, Cout).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    # input gradient = the mirrored kernel on dY with the [ci][tap][co] weights
    if Cout & (Cout - 1) == 0 and Cout >= 8:
        dy = rnd(4, (B, Cout, H, W)).bfloat16().float()
        dyn = torch.empty((B * H * W, Cout), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib().odw_nchw_f32_to_nhwc_bf16(L.ptr(dy), B, H * W, Cout, Cout, L.ptr(dyn), L.stream()), "to nhwc")
        dx = torch.empty((B * H * W, Cin), dtype=torch.float32, device="cuda")
        L.check(L.lib().odw_conv3x3_nhwc_bf16(L.ptr(dyn), B * H * W, H, W, Cout, dil, 1, L.ptr(wd), wd.stride(0), Cin, L.ptr(dx), Cin, 0,
                                              None, 0, None, 0, L.ptr(zero), L.stream()), "dgrad")
        xr = x.clone().requires_grad_(True)
        F.conv2d(xr, w, None, padding=dil, dilation=dil).backward(dy)
        gotdx = dx.reshape(B, H, W, Cin).permute(0, 3, 1, 2)
        assert (gotdx - xr.grad).abs().max().item() <= 2e-3 * max(1.0, xr.grad.abs().max().item())


@pytest.mark.parametrize("B,Cin,Cout,H,W,dil,forced", [(1, 512, 512, 12, 10, 2, 3), (1, 256, 512, 19, 13, 1, 2),
                                                       (1, 512, 512, 76, 76, 2, None), (2, 128, 128, 9, 7, 1, 2)])
def test_conv3x3_split_k_ring(lib, B, Cin, Cout, H, W, dil, forced, monkeypatch):
    """The 256x128 ring form with K split over the grid + the reduction pass (bias, ReLU, ReLU-backward mask), bf16 and
    fp32 outputs, against the unsplit 128x128 kernel and the PyTorch convolution."""
    L = lib
    if forced:
        monkeypatch.setenv("ODW_CONV_SPLITK", str(forced))
    m = B * H * W
    x = rnd(11, (B, Cin, H, W)).bfloat16().float()
    w = rnd(12, (Cout, Cin, 3, 3), 0.03).bfloat16().float()
    b = rnd(13, (Cout,), 0.1)
    zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    xn = torch.empty((m, Cin), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_nchw_f32_to_nhwc_bf16(L.ptr(x), B, H * W, Cin, Cin, L.ptr(xn), L.stream()), "to nhwc")
    wk = torch.empty((Cout, r64(9 * Cin)), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_conv_weight_prep(L.ptr(w), Cout, Cin, Cin, L.ptr(wk), wk.stride(0), None, 0, L.stream()), "prep")
    mask = (rnd(14, (m, Cout)) > 0).to(torch.bfloat16)
    ws_bytes = L.lib().odw_conv3x3_workspace(m, Cin, Cout)
    assert ws_bytes > 0 and ws_bytes % (m * Cout * 4) == 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    ref = torch.relu(F.conv2d(x, w, b, padding=dil, dilation=dil)).permute(0, 2, 3, 1).reshape(m, Cout) * mask.float()
    for dt, tol in ((torch.float32, 2e-3), (torch.bfloat16, 1e-2)):
        y = torch.empty((m, Cout), dtype=dt, device="cuda")
        L.check(L.lib().odw_conv3x3_nhwc_bf16_ws(L.ptr(xn), m, H, W, Cin, dil, 0, L.ptr(wk), wk.stride(0), Cout, L.ptr(y), Cout,
                                                 1 if dt == torch.bfloat16 else 0, L.ptr(b), 1, L.ptr(mask), Cout, L.ptr(zero),
                                                 L.ptr(ws), ws_bytes, L.stream()), "conv ws")
        y0 = torch.empty((m, Cout), dtype=dt, device="cuda")
        L.check(L.lib().odw_conv3x3_nhwc_bf16(L.ptr(xn), m, H, W, Cin, dil, 0, L.ptr(wk), wk.stride(0), Cout, L.ptr(y0), Cout,
                                              1 if dt == torch.bfloat16 else 0, L.ptr(b), 1, L.ptr(mask), Cout, L.ptr(zero),
                                              L.stream()), "conv")
        scale = max(1.0, ref.abs().max().item())
        assert (y.float() - ref).abs().max().item() <= tol * scale
        assert (y.float() - y0.float()).abs().max().item() <= tol * scale


@pytest.mark.parametrize("B,Cin,Cout,H,W,dil,mirror,splits", [
    (2, 64, 64, 37, 45, 1, 0, 1), (1, 128, 192, 33, 16, 2, 0, 2), (1, 256, 64, 17, 50, 2, 1, 4), (3, 64, 128, 16, 16, 1, 1, 1),
    (1, 512, 512, 76, 76, 1, 0, 0), (1, 512, 256, 40, 30, 2, 1, 0), (1, 256, 136, 21, 19, 1, 0, 3)])
def test_conv3x3_halo_tile_kernel(lib, B, Cin, Cout, H, W, dil, mirror, splits, monkeypatch):
    """conv3x3_halo_kernel (16x16 spatial tiles, the input patch staged once per 64-channel block) against PyTorch and
    against the 128x128 kernel on the same operands: ragged image edges, several images, both dilations, mirrored taps
    (input gradient), channel-block slices + reduction pass, N not a multiple of the tile."""
    L = lib
    if splits:
        monkeypatch.setenv("ODW_CONV_SPLITK", str(splits))
    m = B * H * W
    x = rnd(21, (B, Cin, H, W)).bfloat16().float()
    w = rnd(22, (Cout, Cin, 3, 3), 0.03).bfloat16().float()
    b = rnd(23, (Cout,), 0.1)
    zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    xn = torch.empty((m, Cin), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_nchw_f32_to_nhwc_bf16(L.ptr(x), B, H * W, Cin, Cin, L.ptr(xn), L.stream()), "to nhwc")
    # the packed [n][tap*C + c] weights; for the mirrored run the same array plays the [ci][tap*Cout + co] copy
    wk = torch.zeros((Cout, r64(9 * Cin)), dtype=torch.bfloat16, device="cuda")
    wk[:, :9 * Cin] = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).bfloat16()
    mask = (rnd(24, (m, Cout)) > -0.5).to(torch.bfloat16)
    if mirror:      # taps flipped: correlation with the 180-degree rotated kernel
        ref = F.conv2d(x, torch.flip(w, dims=(2, 3)), b, padding=dil, dilation=dil)
    else:
        ref = F.conv2d(x, w, b, padding=dil, dilation=dil)
    ref = torch.relu(ref).permute(0, 2, 3, 1).reshape(m, Cout) * mask.float()
    ws_bytes = L.lib().odw_conv3x3_workspace_hw(m, H, W, Cin, Cout, dil)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
    outs = {}
    for halo in ("1", "0"):
        monkeypatch.setenv("ODW_CONV_HALO", halo)
        for dt in (torch.float32, torch.bfloat16):
            y = torch.full((m, Cout), float("nan"), dtype=dt, device="cuda")
            L.check(L.lib().odw_conv3x3_nhwc_bf16_ws(L.ptr(xn), m, H, W, Cin, dil, mirror, L.ptr(wk), wk.stride(0), Cout, L.ptr(y),
                                                     Cout, 1 if dt == torch.bfloat16 else 0, L.ptr(b), 1, L.ptr(mask), Cout,
                                                     L.ptr(zero), L.ptr(ws) if ws_bytes else None, ws_bytes, L.stream()), "conv")
            outs[(halo, dt)] = y.float()
    scale = max(1.0, ref.abs().max().item())
    for (halo, dt), y in outs.items():
        tol = 2e-3 if dt == torch.float32 else 1e-2
        assert torch.isfinite(y).all(), (halo, dt)
        assert (y - ref).abs().max().item() <= tol * scale, (halo, dt)
    # same products, same fp32 accumulation up to the order of the K walk
    assert (outs[("1", torch.float32)] - outs[("0", torch.float32)]).abs().max().item() <= 2e-4 * scale


def test_maxpool_and_layout_kernels(lib):
    L = lib
    B, C, H, W = 2, 16, 8, 12
    x = torch.relu(rnd(5, (B, C, H, W))).bfloat16().float()
    x[:, :, 0:2, 0:2] = 0                                   # an all-zero window: first element wins, ReLU mask kills it
    xn = torch.empty((B * H * W, C), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_nchw_f32_to_nhwc_bf16(L.ptr(x), B, H * W, C, C, L.ptr(xn), L.stream()), "to nhwc")
    back = torch.empty((B, C, H, W), device="cuda")
    L.check(L.lib().odw_nhwc_bf16_to_nchw_f32(L.ptr(xn), B, H * W, C, L.ptr(back), L.stream()), "to nchw")
    assert torch.equal(back, x)
    p = torch.empty((B * (H // 2) * (W // 2), C), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_maxpool2x2_nhwc_bf16(L.ptr(xn), B, H, W, C, L.ptr(p), L.stream()), "pool")
    ref = F.max_pool2d(x, 2, 2)
    assert torch.equal(p.float().reshape(B, H // 2, W // 2, C).permute(0, 3, 1, 2), ref)
    dy = rnd(6, ref.shape).bfloat16().float()
    dyn = torch.empty((B * (H // 2) * (W // 2), C), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_nchw_f32_to_nhwc_bf16(L.ptr(dy), B, (H // 2) * (W // 2), C, C, L.ptr(dyn), L.stream()), "to nhwc")
    dx = torch.empty((B * H * W, C), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_maxpool2x2_nhwc_bf16_bwd(L.ptr(xn), L.ptr(dyn), B, H, W, C, L.ptr(dx), L.stream()), "pool bwd")
    xr = x.clone().requires_grad_(True)
    pre = torch.relu(xr)                                    # pooled activation is post-ReLU: mask folded into the kernel
    F.max_pool2d(pre, 2, 2).backward(dy)
    assert torch.equal(dx.float().reshape(B, H, W, C).permute(0, 3, 1, 2), xr.grad)


def test_vgg16_backbone_forward_backward(lib):
    """Whole backbone, forward + backward, vs the torch fp32 model holding the same (bf16-rounded) weights."""
    from od_wscl_amd.config import make_defaults
    from od_wscl_amd.modeling.backbone import build_backbone
    from od_wscl_amd.modeling.backbone.vgg16_hip import VGGBackboneHip
    cfg = make_defaults()
    cfg.merge_from_list(["MODEL.BACKBONE.CONV_BODY", "VGG16-OICR"])
    torch.manual_seed(0)
    bb = build_backbone(cfg).cuda()
    with torch.no_grad():
        for p in bb.parameters():
            p.copy_(p.bfloat16().float())
            if p.dim() == 1:
                p.normal_(0, 0.05)
    hip = VGGBackboneHip(bb.body)
    x = rnd(7, (1, 3, 64, 96), 50.0)
    feat = hip(x)[0]
    g = rnd(8, tuple(feat.shape))
    feat.backward(g)
    got = {n: p.grad.clone() for n, p in bb.named_parameters() if p.grad is not None}
    for p in bb.parameters():
        p.grad = None
    # reference: the same torch modules in fp32, activations rounded to bf16 where the kernels store bf16
    # (otherwise ~0.3 % of the ReLU masks differ from pure-fp32 activations, which alone moves the weight
    # gradients by ~5 % in L2 -- that is bf16 storage, not the kernels)
    class RoundBF16(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.bfloat16().float()

        @staticmethod
        def backward(ctx, gr):
            return gr.bfloat16().float()
    # ... and on the CPU: MIOpen's fp32 convolutions (Winograd) are only ~1e-3 accurate, which flips ~0.1 % of
    # the ReLU masks by itself
    got = {n: t.cpu() for n, t in got.items()}
    feat, g = feat.detach().cpu(), g.cpu()
    bb = bb.cpu()
    h = x.bfloat16().float().cpu()
    from od_wscl_amd.layers.misc import library_reference
    for m in bb.body.features:
        with library_reference():                  # torch's fp32 convolution on the CPU: the reference, not the product
            h = m(h)
        if isinstance(m, (torch.nn.ReLU, torch.nn.MaxPool2d)) or m is bb.body.features[-1]:
            h = RoundBF16.apply(h)
    ref = h
    ref.backward(g)
    cos = lambda a, b: (a.flatten() @ b.flatten() / (a.norm() * b.norm() + 1e-30)).item()
    assert feat.shape == ref.shape
    assert cos(feat, ref) > 0.999 and abs(feat.norm().item() / ref.norm().item() - 1) < 2e-2
    names = [n for n, p in bb.named_parameters() if p.requires_grad]
    assert set(got) == set(names)                            # frozen conv1_x / conv2_x get no gradient
    report = {n: (round(cos(got[n], p.grad), 5), round(got[n].norm().item() / p.grad.norm().item(), 4))
              for n, p in bb.named_parameters() if p.requires_grad}
    print("GRADREPORT", " ".join("%s:%s/%s" % (n.replace("body.features.", "f"), c, r) for n, (c, r) in report.items()))
    # Two bf16-storage pipelines with different fp32 summation orders diverge chaotically (a 1-ulp bf16 flip in
    # layer k perturbs layer k+1, ~0.15 % of the conv5 ReLU masks differ after 12 layers), so the end-to-end bar is
    # loose; the tight bars are the per-operator tests above and test_conv_weight_gradient below.
    for n, (c, r) in report.items():
        assert c > 0.98 and abs(r - 1) < 5e-2, (n, c, r)
    assert report["body.features.28.weight"][0] > 0.9999


@pytest.mark.parametrize("B,Cin,Cout,H,W,dil", [(1, 64, 128, 16, 24, 1), (2, 256, 256, 9, 11, 2), (1, 128, 64, 38, 50, 1),
                                                  (1, 512, 512, 19, 13, 2), (1, 256, 264, 76, 76, 1)])
def test_conv_weight_gradient(lib, B, Cin, Cout, H, W, dil):
    """bias + weight gradient of one layer: transposed im2col + NT GEMM + unpack vs autograd."""
    from od_wscl_amd import gemm
    L = lib
    m = B * H * W
    m64 = r64(m)
    x = rnd(11, (B, Cin, H, W)).bfloat16().float()
    dy = rnd(12, (B, Cout, H, W)).bfloat16().float()
    xn = torch.empty((m, Cin), dtype=torch.bfloat16, device="cuda")
    dyn = torch.empty((m, Cout), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_nchw_f32_to_nhwc_bf16(L.ptr(x), B, H * W, Cin, Cin, L.ptr(xn), L.stream()), "to nhwc")
    L.check(L.lib().odw_nchw_f32_to_nhwc_bf16(L.ptr(dy), B, H * W, Cout, Cout, L.ptr(dyn), L.stream()), "to nhwc")
    dzc = torch.empty((m, r64(Cout)), dtype=torch.bfloat16, device="cuda")
    dzt = torch.empty((Cout, m64), dtype=torch.bfloat16, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    L.check(L.lib().odw_linear_bwd_prep(L.ptr(dyn), 0, Cout, None, 0, m, Cout, 1.0, L.ptr(dzc), dzc.stride(0), L.ptr(dzt), m64,
                                        L.ptr(db), L.stream()), "prep")
    colt = torch.empty((9 * Cin, m64), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_im2col_t_bf16(L.ptr(xn), m, H, W, Cin, dil, L.ptr(colt), m64, L.stream()), "im2col_t")
    dwk = torch.empty((Cout, 9 * Cin), device="cuda")
    gemm.gemm_nt(dzt, colt, Cout, 9 * Cin, m, dwk)
    dw = torch.empty((Cout, Cin, 3, 3), device="cuda")
    L.check(L.lib().odw_conv_wgrad_unpack(L.ptr(dwk), 9 * Cin, Cout, Cin, Cin, L.ptr(dw), L.stream()), "unpack")
    w = torch.zeros((Cout, Cin, 3, 3), device="cuda", requires_grad=True)
    b = torch.zeros(Cout, device="cuda", requires_grad=True)
    F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=dil, dilation=dil).backward(dy.cpu())   # CPU: exact fp32 reference
    wg = torch.autograd.grad(F.conv2d(x, w, b, padding=dil, dilation=dil), [w, b], dy)
    ref_w, ref_b = wg[0], wg[1]
    assert (dw - ref_w).abs().max().item() <= 3e-3 * max(1.0, ref_w.abs().max().item())
    assert (db - ref_b).abs().max().item() <= 3e-3 * max(1.0, ref_b.abs().max().item())
    # the one-call form (split-K partials + ONE reduce-and-unpack pass): the same sums in the same slice order
    ws_bytes = L.lib().odw_conv_wgrad_workspace(Cout, Cin, m, dzt.stride(0), colt.stride(0))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    dw1 = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    L.check(L.lib().odw_conv_wgrad_nt(L.ptr(dzt), dzt.stride(0), L.ptr(colt), colt.stride(0), Cout, Cin, Cin, m, L.ptr(dw1), 0,
                                      L.ptr(ws), ws_bytes, L.stream()), "conv_wgrad_nt")
    assert torch.equal(dw1, dw)
    L.check(L.lib().odw_conv_wgrad_nt(L.ptr(dzt), dzt.stride(0), L.ptr(colt), colt.stride(0), Cout, Cin, Cin, m, L.ptr(dw1), 1,
                                      L.ptr(ws), ws_bytes, L.stream()), "conv_wgrad_nt accumulate")
    assert torch.equal(dw1, dw + dw)
    # the TN form: the same gradient straight from the NHWC operands (K-major LDS tiles, transposed fragment reads)
    if Cin >= 128 and Cin & (Cin - 1) == 0:
        zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
        wsb = L.lib().odw_conv_wgrad_tn_workspace(Cout, Cin, m)
        ws2 = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        dw2 = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
        L.check(L.lib().odw_conv_wgrad_tn(L.ptr(dyn), dyn.stride(0), L.ptr(xn), m, H, W, Cin, dil, Cout, Cin, L.ptr(dw2), 0,
                                          L.ptr(zero), L.ptr(ws2), wsb, L.stream()), "conv_wgrad_tn")
        scale = max(1.0, ref_w.abs().max().item())
        assert torch.isfinite(dw2).all()
        assert (dw2 - ref_w).abs().max().item() <= 3e-3 * scale
        assert (dw2 - dw).abs().max().item() <= 2e-4 * scale       # same bf16 products, another summation order
        db2 = torch.zeros(Cout, device="cuda")
        L.check(L.lib().odw_colsum_bf16(L.ptr(dyn), dyn.stride(0), m, Cout, L.ptr(db2), L.stream()), "colsum")
        assert (db2 - ref_b).abs().max().item() <= 3e-3 * max(1.0, ref_b.abs().max().item())
        # the two-stage form the training step uses: same sums, ONE summation order -> identical from run to run, added
        # onto what `out` holds
        csb = L.lib().odw_colsum_workspace(m, Cout)
        cws = torch.empty(csb, dtype=torch.uint8, device="cuda")
        outs = []
        for _ in range(3):
            db3 = torch.full((Cout,), 0.5, device="cuda")
            L.check(L.lib().odw_colsum_bf16_ws(L.ptr(dyn), dyn.stride(0), m, Cout, L.ptr(db3), L.ptr(cws), csb, L.stream()), "colsum_ws")
            outs.append(db3 - 0.5)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
        assert (outs[0] - ref_b).abs().max().item() <= 3e-3 * max(1.0, ref_b.abs().max().item())
        # weight AND bias gradient from one call (channel counts that are multiples of 64: extra workgroups of the halo
        # kernel form the column sums on the matrix pipe; otherwise ring form + two-launch column sum): the same dw bits as
        # odw_conv_wgrad_tn, db added onto what it holds, identical from run to run
        wsb3 = L.lib().odw_conv_wgrad_tn_bias_workspace(Cout, Cin, m)
        ws3 = torch.empty(wsb3, dtype=torch.uint8, device="cuda")
        dbs = []
        for _ in range(3):
            dw3 = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
            db4 = torch.full((Cout,), 0.25, device="cuda")
            L.check(L.lib().odw_conv_wgrad_tn_bias(L.ptr(dyn), dyn.stride(0), L.ptr(xn), xn.stride(0), m, H, W, Cin, dil, Cout, Cin,
                                                   L.ptr(dw3), L.ptr(db4), 0, L.ptr(zero), L.ptr(ws3), wsb3, L.stream()), "conv_wgrad_tn_bias")
            assert torch.equal(dw3, dw2)
            dbs.append(db4 - 0.25)
        assert torch.equal(dbs[0], dbs[1]) and torch.equal(dbs[1], dbs[2])
        assert (dbs[0] - ref_b).abs().max().item() <= 3e-3 * max(1.0, ref_b.abs().max().item())
        if Cout % 64 == 0 and Cin % 64 == 0:
            # X as the first block of a wider planes operand, read in place (row stride 3 Cin): the same bits
            wide = torch.full((m, 3 * Cin), 9.0, dtype=torch.bfloat16, device="cuda")
            wide[:, :Cin] = xn
            dw5 = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
            db5 = torch.zeros(Cout, device="cuda")
            L.check(L.lib().odw_conv_wgrad_tn_bias(L.ptr(dyn), dyn.stride(0), L.ptr(wide), wide.stride(0), m, H, W, Cin, dil, Cout,
                                                   Cin, L.ptr(dw5), L.ptr(db5), 0, L.ptr(zero), L.ptr(ws3), wsb3, L.stream()), "strided X")
            assert torch.equal(dw5, dw2) and torch.equal(db5, dbs[0])


def test_conv_weight_prep_batch_equals_per_layer(lib):
    """odw_conv_weight_prep_batch (every layer in one launch, coalesced through LDS) writes exactly what the per-layer
    kernel writes, zero padding included: both packed layouts, a padded channel count, a layer without the mirrored copy."""
    import ctypes
    L = lib
    shapes = [(64, 3, 8, True), (128, 64, 64, True), (520, 256, 256, False), (256, 512, 512, True), (72, 40, 64, True)]
    ws, wk0, wd0, wk1, wd1 = [], [], [], [], []
    for i, (co, ci, cp, mirrored) in enumerate(shapes):
        w = rnd(40 + i, (co, ci, 3, 3), 0.1)
        ws.append(w)
        for wk, wd in ((wk0, wd0), (wk1, wd1)):
            wk.append(torch.full((co, r64(9 * cp)), 7.0, dtype=torch.bfloat16, device="cuda"))
            wd.append(torch.full((ci, r64(9 * co)), 7.0, dtype=torch.bfloat16, device="cuda") if mirrored else None)
        L.check(L.lib().odw_conv_weight_prep(L.ptr(w), co, ci, cp, L.ptr(wk0[i]), wk0[i].stride(0), L.ptr(wd0[i]),
                                             wd0[i].stride(0) if mirrored else 0, L.stream()), "prep")
    n = len(shapes)
    vp, ia = ctypes.c_void_p * n, ctypes.c_int * n
    args = (vp(*[w.data_ptr() for w in ws]), ia(*[s[0] for s in shapes]), ia(*[s[1] for s in shapes]), ia(*[s[2] for s in shapes]),
            vp(*[t.data_ptr() for t in wk1]), ia(*[t.stride(0) for t in wk1]),
            vp(*[t.data_ptr() if t is not None else None for t in wd1]), ia(*[t.stride(0) if t is not None else 0 for t in wd1]))
    L.check(L.lib().odw_conv_weight_prep_batch(n, *[ctypes.cast(a, ctypes.c_void_p) for a in args], L.stream()), "prep batch")
    for i in range(n):
        assert torch.equal(wk0[i].view(torch.int16), wk1[i].view(torch.int16)), i
        if wd0[i] is not None:
            assert torch.equal(wd0[i].view(torch.int16), wd1[i].view(torch.int16)), i


def test_conv_weight_prep_planes_batch_equals_the_split_of_the_packed_weight(lib):
    """odw_conv_weight_prep_planes_batch ("bf16x2f": the forward operand of every convolution as bf16 planes, all layers
    in one launch) writes exactly what precision.pack_conv_weight builds from torch passes -- per tap T blocks of Cp
    channels holding plane pattern[t], rows zero padded to a multiple of 64 -- and the plain bf16 mirrored copy beside it."""
    import ctypes
    from od_wscl_amd import precision as P
    L = lib
    shapes = [(64, 64, 64, (0, 1, 0)), (256, 128, 128, (0, 1, 0)), (72, 40, 64, (0, 1, 2, 0)), (64, 3, 8, (0, 1, 0, 1)),
              (512, 256, 256, (0, 1, 0)), (96, 192, 192, (0, 1, 2, 0))]      # the last two: the tiled form
    ws, wk, wd, want = [], [], [], []
    for i, (co, ci, cp, pat) in enumerate(shapes):
        w = rnd(60 + i, (co, ci, 3, 3), 0.1)
        ws.append(w)
        T = len(pat)
        wk.append(torch.full((co, r64(9 * T * cp)), 7.0, dtype=torch.bfloat16, device="cuda"))
        wd.append(torch.full((ci, r64(9 * co)), 7.0, dtype=torch.bfloat16, device="cuda"))
        wr = torch.zeros((co, 9, cp), dtype=torch.float32, device="cuda")
        wr[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, 9, ci)
        want.append(P.pack_conv_weight(wr.view(co * 9, cp), pat, cp, co))
    n = len(shapes)
    vp, ia = ctypes.c_void_p * n, ctypes.c_int * n
    pats = [list(s[3]) for s in shapes]
    args = (vp(*[w.data_ptr() for w in ws]), ia(*[s[0] for s in shapes]), ia(*[s[1] for s in shapes]), ia(*[s[2] for s in shapes]),
            vp(*[t.data_ptr() for t in wk]), ia(*[t.stride(0) for t in wk]), vp(*[t.data_ptr() for t in wd]),
            ia(*[t.stride(0) for t in wd]), ia(*[len(pt) for pt in pats]),
            (ctypes.c_int * (4 * n))(*[v for pt in pats for v in (pt + [3] * 4)[:4]]))
    L.check(L.lib().odw_conv_weight_prep_planes_batch(n, *[ctypes.cast(a, ctypes.c_void_p) for a in args], L.stream()), "prep planes")
    for i, (co, ci, cp, pat) in enumerate(shapes):
        assert want[i].shape == wk[i].shape, (want[i].shape, wk[i].shape)
        assert torch.equal(want[i].view(torch.int16), wk[i].view(torch.int16)), i
        plain = torch.empty_like(wd[i])
        L.check(L.lib().odw_conv_weight_prep(L.ptr(ws[i]), co, ci, cp, None, 0, L.ptr(plain), plain.stride(0), L.stream()), "prep")
        assert torch.equal(plain.view(torch.int16), wd[i].view(torch.int16)), i


@pytest.mark.parametrize("hw", [(37, 50), (64, 64)])
def test_stem_planes64_kernel_equals_the_per_pixel_form(lib, hw):
    """odw_stem_conv3x3_bias_relu_planes with Co = 64 runs the register-resident-weights kernel (16 lanes per pixel, runs of
    four pixels); with any other channel count the one-thread-per-pixel kernel.  Both evaluate the same fmaf chain per
    output: the 64 channels of the first must equal, bit for bit, the first 64 of a 72-channel launch of the second
    (same weights), plane block by plane block, at image sizes that are not multiples of the run length."""
    import ctypes
    L = lib
    H, W = hw
    B = 2
    img = rnd(70, (B, 3, H, W), 1.0)
    w72 = rnd(71, (72, 3, 3, 3), 0.2)
    b72 = rnd(72, (72,), 0.1)
    pat = (ctypes.c_int * 3)(0, 0, 1)
    out = {}
    for co, blk in ((64, 64), (72, 72)):
        o = torch.full((B * H * W, 3 * blk), 7.0, dtype=torch.bfloat16, device="cuda")
        L.check(L.lib().odw_stem_conv3x3_bias_relu_planes(L.ptr(img), L.ptr(w72[:co].contiguous()), L.ptr(b72[:co].contiguous()),
                                                          B, H, W, co, ctypes.cast(pat, ctypes.c_void_p), 3, L.ptr(o),
                                                          o.stride(0), blk, L.stream()), "stem planes")
        out[co] = o.view(B * H * W, 3, blk)[:, :, :64].contiguous()
    assert torch.equal(out[64].view(torch.int16), out[72].view(torch.int16))
    # and it is the convolution: hi + mid of the planes against torch's fp32 conv + ReLU
    ref = torch.relu(torch.nn.functional.conv2d(img, w72[:64], b72[:64], padding=1)).permute(0, 2, 3, 1).reshape(B * H * W, 64)
    got = out[64][:, 0].float() + out[64][:, 2].float()
    assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
