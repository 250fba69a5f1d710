"""The two-plane forward convolution of the "bf16x2f" mode (csrc/gemm_bf16.hip: conv3x3_halo2_kernel; round 5) against fp64.

The reference computes its convolutions in fp32 (modeling/backbone/vgg16.py:34-36,58-83 on cuDNN, config/defaults.py:559).
The timed mode forms each as x_hi w_hi + x_hi w_mid + x_mid w_hi with fp32 accumulation, where hi = bf16(v) and
mid = bf16(v - hi): every term dropped is <= 2^-16 of a product.  Checked here, through the C-ABI:
  * the operand formats: planes [hi C | mid C] per pixel in, weights packed per tap and block of 32 channels as
    [hi 32 | mid 32] (odw_conv_weight_prep_planes_batch, T = -2), fp32 or planes out;
  * values against the fp64 convolution of the SAME fp32 inputs at 2^-14 of the output scale (a single-plane bf16
    product is ~2^-8 off: the mid planes must have taken part), every tile shape: 64 / 128 / 256+ output channels, dilation
    1 and 2, maps that are not multiples of the 16 x 16 tile, several images, the K-sliced form of small maps;
  * the planes output == the split of the fp32 output, bit for bit (same accumulators, same split);
  * the fused pooling kernel == max pool + split.
And the body: the bf16x2f forward of the whole VGG16 through the two-plane path equals the three-pass path of rounds 3-4
(ODW_CONV_PLANES2=0) to fp32 re-association, with gradients flowing through the same backward."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from od_wscl_amd.utils import rng

pytestmark = pytest.mark.gpu


def rnd(seed, shape, scale=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy((rng.normal(seed, 1, n) * scale).reshape(shape)).cuda()


def r64(n):
    return (n + 63) // 64 * 64


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import _lib
    return _lib


def split_planes(v):
    """fp32 -> (hi, mid) bf16 tensors, round to nearest even twice (what odw_planes.h: split2 does)."""
    hi = v.bfloat16()
    mid = (v - hi.float()).bfloat16()
    return hi, mid


def nhwc(t):
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def pack_weights(L, w, cp):
    cout, cin = w.shape[:2]
    wk = torch.full((cout, r64(18 * cp)), 7.0, dtype=torch.bfloat16, device="cuda")
    vp, ia = ctypes.c_void_p * 1, ctypes.c_int * 1
    args = [vp(w.data_ptr()), ia(cout), ia(cin), ia(cp), vp(wk.data_ptr()), ia(wk.stride(0)), vp(None), ia(0), ia(-2),
            (ctypes.c_int * 4)(3, 3, 3, 3)]
    L.check(L.lib().odw_conv_weight_prep_planes_batch(1, *[ctypes.cast(a, ctypes.c_void_p) for a in args], L.stream()), "prep")
    return wk


def run_conv(L, x, w, b, dil, relu, planes_out):
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    m = B * H * W
    hi, mid = split_planes(nhwc(x))
    xs = torch.cat([hi, mid], dim=1).contiguous()                       # (m, 2 Cin): [hi | mid]
    wk = pack_weights(L, w, Cin)
    zero = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    ws_bytes = L.lib().odw_conv3x3_planes2_workspace(m, H, W, Cin, Cout)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
    if planes_out == 2:
        y = torch.full((B * (H // 2) * (W // 2), 2 * Cout), 5.0, dtype=torch.bfloat16, device="cuda")
    elif planes_out:
        y = torch.full((m, 2 * Cout), 5.0, dtype=torch.bfloat16, device="cuda")
    else:
        y = torch.full((m, Cout), 5.0, dtype=torch.float32, device="cuda")
    L.check(L.lib().odw_conv3x3_planes2_ws(L.ptr(xs), xs.stride(0), m, H, W, Cin, dil, L.ptr(wk), wk.stride(0), Cout, L.ptr(y),
                                           y.stride(0), int(planes_out), L.ptr(b), 1 if relu else 0, L.ptr(zero),
                                           L.ptr(ws) if ws_bytes else None, ws_bytes, L.stream()), "conv3x3_planes2")
    return y, ws_bytes


CASES = [  # B, Cin, Cout, H, W, dilation, relu
    (1, 64, 64, 40, 36, 1, True),        # the 8 x 1 wave layout of a 64-channel layer (conv1_2)
    (2, 64, 128, 19, 23, 1, True),       # two images, ragged tiles
    (1, 128, 256, 33, 17, 1, False),     # no ReLU (the last layer of the body has none, vgg16.py:82-83)
    (1, 256, 256, 16, 16, 2, True),      # dilation 2, K sliced (one tile)
    (1, 512, 512, 12, 10, 2, True),      # conv5_x shape, K sliced
    (1, 32, 64, 8, 8, 1, True),          # one channel block
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,dil,relu", CASES)
def test_two_plane_convolution_matches_fp64(L, B, Cin, Cout, H, W, dil, relu):
    x = rnd(11, (B, Cin, H, W))
    w = rnd(12, (Cout, Cin, 3, 3), 0.05)
    b = rnd(13, (Cout,), 0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=dil, dilation=dil)
    if relu:
        ref = torch.relu(ref)
    ref = nhwc(ref)
    scale = max(1.0, ref.abs().max().item())
    y, ws_bytes = run_conv(L, x, w, b, dil, relu, False)
    err = (y.double() - ref).abs().max().item()
    assert err <= 2.0 ** -14 * scale, (err, scale)
    # the single-plane product would NOT pass: the bar above tests that the mid planes took part
    one = F.conv2d(x.bfloat16().double(), w.bfloat16().double(), b.double(), padding=dil, dilation=dil)
    one = nhwc(torch.relu(one) if relu else one)
    assert (one - ref).abs().max().item() > 2.0 ** -12 * scale
    # planes output: the split of the same accumulators
    yp, _ = run_conv(L, x, w, b, dil, relu, True)
    hi, mid = split_planes(y)
    assert torch.equal(yp[:, :Cout], hi) and torch.equal(yp[:, Cout:], mid)
    # K slices (forced) give the same sums up to fp32 re-association
    for k in ("1", "2"):
        os.environ["ODW_CONV_SPLITK"] = k
        try:
            yk, _ = run_conv(L, x, w, b, dil, relu, False)
            ykp, _ = run_conv(L, x, w, b, dil, relu, True)
        finally:
            os.environ.pop("ODW_CONV_SPLITK", None)
        assert (yk.double() - ref).abs().max().item() <= 2.0 ** -14 * scale
        hk, mk = split_planes(yk)
        assert torch.equal(ykp[:, :Cout], hk) and torch.equal(ykp[:, Cout:], mk)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(1, 64, 64, 40, 36), (2, 64, 128, 18, 22), (1, 128, 128, 34, 50)])
def test_pool_and_split_in_the_convolution_epilogue(L, B, Cin, Cout, H, W, monkeypatch):
    """y_planes = 2: bias + ReLU + 2 x 2 max pool + split in the epilogue == the fp32 output pooled and split afterwards, bit for
    bit (the same accumulators; max and split are exact operations).  The pooled form never slices K, so the fp32 side is run
    unsliced too (a sliced sum differs in its last bit); the second half runs the pooled form where the plan WOULD slice."""
    x = rnd(41, (B, Cin, H, W))
    w = rnd(42, (Cout, Cin, 3, 3), 0.05)
    b = rnd(43, (Cout,), 0.1)
    yp_sliced_plan, _ = run_conv(L, x, w, b, 1, True, 2)        # these shapes have < 256 tiles: the plan asks for K slices
    monkeypatch.setenv("ODW_CONV_SPLITK", "1")
    y, _ = run_conv(L, x, w, b, 1, True, False)
    pooled = F.max_pool2d(y.view(B, H, W, Cout).permute(0, 3, 1, 2), 2)
    hi, mid = split_planes(nhwc(pooled))
    yp, _ = run_conv(L, x, w, b, 1, True, 2)
    assert yp.shape == (B * (H // 2) * (W // 2), 2 * Cout)
    assert torch.equal(yp[:, :Cout], hi) and torch.equal(yp[:, Cout:], mid)
    assert torch.equal(yp_sliced_plan, yp)


@pytest.mark.parametrize("B,Cin,Cout,H,W,planes_out", [(1, 128, 128, 40, 36, 0), (2, 64, 128, 19, 23, 1), (1, 96, 256, 33, 17, 1),
                                                        (1, 128, 128, 34, 50, 2), (1, 32, 128, 24, 24, 0)])
def test_two_steps_per_barrier_equal_one_step_per_barrier(L, B, Cin, Cout, H, W, planes_out, monkeypatch):
    """The PAIR form of the dilation-1 kernel (two K steps per workgroup barrier on a four-slot weight ring: the default) walks the
    same K steps in the same order into the same accumulators as the one-step form (ODW_CONV_PAIRSTEP=0): identical bits, for an
    even and an odd number of steps (9 x channel blocks), with and without K slices, for all three output forms."""
    x = rnd(51, (B, Cin, H, W))
    w = rnd(52, (Cout, Cin, 3, 3), 0.05)
    b = rnd(53, (Cout,), 0.1)
    for forced in (None, "1", "2"):
        if forced is None:
            monkeypatch.delenv("ODW_CONV_SPLITK", raising=False)
        else:
            monkeypatch.setenv("ODW_CONV_SPLITK", forced)
        monkeypatch.delenv("ODW_CONV_PAIRSTEP", raising=False)
        y_pair, _ = run_conv(L, x, w, b, 1, True, planes_out)
        monkeypatch.setenv("ODW_CONV_PAIRSTEP", "0")
        y_one, _ = run_conv(L, x, w, b, 1, True, planes_out)
        assert torch.equal(y_pair.view(torch.int16) if y_pair.dtype == torch.bfloat16 else y_pair.view(torch.int32),
                           y_one.view(torch.int16) if y_one.dtype == torch.bfloat16 else y_one.view(torch.int32)), forced


def test_weight_layout_is_hi_mid_per_block_of_32_channels(L):
    cout, cin = 64, 96
    w = rnd(21, (cout, cin, 3, 3), 0.05)
    wk = pack_weights(L, w, cin).float()
    hi, mid = split_planes(w)
    for co, t, ci in ((0, 0, 0), (5, 4, 31), (63, 8, 32), (17, 2, 95), (40, 7, 64)):
        k = t * 2 * cin + (ci // 32) * 64 + ci % 32
        assert wk[co, k].item() == hi[co, ci, t // 3, t % 3].float().item()
        assert wk[co, k + 32].item() == mid[co, ci, t // 3, t % 3].float().item()
    assert (wk[:, 18 * cin:] == 0).all()                      # the pad of a row (to a multiple of 64) is zeroed


@pytest.mark.parametrize("B,C,H,W", [(1, 64, 16, 24), (2, 256, 6, 10)])
def test_pool_and_split_in_one_pass(L, B, C, H, W):
    x = rnd(31, (B * H * W, C))
    y = torch.full((B * (H // 2) * (W // 2), 2 * C), 5.0, dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().odw_maxpool2x2_nhwc_f32_planes2(L.ptr(x), B, H, W, C, L.ptr(y), y.stride(0), L.stream()), "pool")
    ref = F.max_pool2d(x.view(B, H, W, C).permute(0, 3, 1, 2), 2)
    hi, mid = split_planes(nhwc(ref))
    assert torch.equal(y[:, :C], hi) and torch.equal(y[:, C:], mid)


def test_body_two_plane_path_equals_the_three_pass_path(monkeypatch):
    """The bf16x2f forward of the whole VGG16-OICR body: two-plane path (default) vs the three-pass path of rounds 3-4.
    The three plane products are the same numbers; only the order of the fp32 accumulation differs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import precision
    from od_wscl_amd.modeling.backbone import build_backbone, vgg16_hip
    from od_wscl_amd.config import make_defaults
    precision.set_precision("bf16x2f")
    cfg = make_defaults()
    cfg.merge_from_list(["MODEL.BACKBONE.CONV_BODY", "VGG16-OICR"])
    base = build_backbone(cfg).body.cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():
        for p in base.parameters():
            if p.dim() == 4:
                fan_out = p.shape[0] * 9
                p.copy_(torch.randn(p.shape, device="cuda", generator=g) * (2.0 / fan_out) ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, device="cuda", generator=g) * 0.1)
    img = torch.randn(1, 3, 96, 128, device="cuda", generator=g) * 60.0
    outs, grads = [], []
    for flag in ("1", "0"):
        monkeypatch.setenv("ODW_CONV_PLANES2", flag)
        net = vgg16_hip.VGGBackboneHip(base)
        assert vgg16_hip.planes2_body(net) == (flag == "1")
        for p in base.parameters():
            p.grad = None
        feat = net(img)[0]
        (feat * torch.linspace(-1, 1, feat.numel(), device="cuda").view_as(feat)).sum().backward()
        outs.append(feat.detach().clone())
        grads.append({n: p.grad.detach().clone() for n, p in base.named_parameters() if p.grad is not None})
    scale = outs[1].abs().max().item()
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * scale
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) >= 18
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        # the same backward kernels on the same kind of operands; the forward values differ by fp32 re-association (1e-5),
        # which flips a few ReLU masks / bf16 roundings of the saved hi planes: measured 3e-3 at most
        assert (a - b).norm().item() <= 1e-2 * b.norm().item() + 1e-12, n
