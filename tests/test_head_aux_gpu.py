"""csrc/head_aux.hip: the stacked clean + DropBlock GEMM operand and its gradient against the straight PyTorch
rendition of the reference's DropBlock2D.forward (modeling/dropblock/drop_block.py:45-50) + flatten + cat + cast."""
import os

import numpy as np
import pytest
import torch

from od_wscl_amd.utils import rng

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _bf16_precision():
    """The stacked-operand kernels are pinned in their bf16 form here (bit-exact against the PyTorch rendition)."""
    from od_wscl_amd import precision
    precision.set_precision("bf16")
    yield


def _inputs(P, C, h, w, seed):
    x = torch.from_numpy(rng.normal(seed, 1, P * C * h * w).reshape(P, C, h, w)).cuda()
    keep = torch.from_numpy((rng.uniform(seed, 2, P * h * w) > 0.2).astype(np.float32).reshape(P, h, w)).cuda()
    return x, keep


@pytest.mark.parametrize("P,C,h,w", [(37, 64, 7, 7), (5, 512, 7, 7), (3, 16, 14, 14), (1, 64, 2, 2)])
def test_stack_clean_aug_forward_backward(P, C, h, w):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd.modeling.backbone.fc_extractor import _StackCleanAug
    x, keep = _inputs(P, C, h, w, 5)
    xs = x.clone().requires_grad_(True)
    out = _StackCleanAug.apply(xs, keep, keep.sum(), None)
    aug = x * keep[:, None] * keep.numel() / keep.sum()                       # drop_block.py:49-50, same order
    ref = torch.cat([x.reshape(P, -1), aug.reshape(P, -1)]).to(torch.bfloat16)
    assert out.dtype == torch.bfloat16 and out.shape == (2 * P, C * h * w)
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))          # bit-exact: same fp32 ops, same rounding
    from od_wscl_amd import _lib as L
    g32 = torch.from_numpy(rng.normal(6, 3, 2 * P * C * h * w).reshape(2 * P, -1)).cuda()
    out.backward(g32)                       # autograd hands the node a gradient in the output's dtype (bf16)
    for g, got in ((g32.to(torch.bfloat16), xs.grad), (g32, None)):
        if got is None:                     # the fp32-gradient entry point of the kernel, called directly
            got = torch.empty_like(x)
            L.check(L.lib().odw_unstack_clean_aug_bwd(L.ptr(g), 1, g.stride(0), L.ptr(keep), L.ptr(keep.sum()), P, C,
                                                      h * w, L.ptr(got), L.stream()), "unstack")
        gf = g.float().reshape(2, P, C, h, w)
        exp = gf[0] + gf[1] * keep[:, None] * keep.numel() / keep.sum()
        np.testing.assert_allclose(got.cpu().numpy(), exp.cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_stacked_path_equals_unfused_path():
    """forward_clean_and_aug: the fused operand path against the cat/DropBlock2D path on the bf16 back end --
    identical random draws, identical bf16 operand, so identical fc outputs and pooled gradients."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd.config import make_defaults
    from od_wscl_amd import precision as ll
    from od_wscl_amd.modeling.backbone import fc_extractor as fx
    from od_wscl_amd.modeling.backbone.vgg16 import VGG16FC67ROIFeatureExtractor
    from od_wscl_amd.utils.device_rand import DeviceRand
    cfg = make_defaults()
    cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.POOLER_METHOD", "ROIPool", "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", 7,
                         "MODEL.ROI_BOX_HEAD.POOLER_SCALES", (0.125,), "DB.METHOD", "dropblock"])
    ll.set_precision("bf16")
    try:
        torch.manual_seed(0)
        fe = VGG16FC67ROIFeatureExtractor(cfg, 512).cuda().train()
        pooled = torch.randn(40, 512, 7, 7, device="cuda").relu_()
        res = []
        for fused in (True, False):
            fe.rand = DeviceRand(77)
            p = pooled.clone().requires_grad_(True)
            os.environ["ODW_NO_STACK_FUSE"] = "0" if fused else "1"      # only steers the branch in forward_clean_and_aug
            try:
                c, a = fe.forward_clean_and_aug(p)
            finally:
                os.environ.pop("ODW_NO_STACK_FUSE", None)
            (c.float().square().sum() + a.float().sum()).backward()
            res.append((c.float(), a.float(), p.grad.clone(), fe.rand.s.next))
        assert res[0][3] == res[1][3]                                  # same number of random draws
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        # the unfused path hands fc6 an fp32 gradient, the fused one a bf16 gradient: 2^-8 relative
        d = (res[0][2] - res[1][2]).abs().max().item()
        assert d <= 1e-2 * res[1][2].abs().max().item(), d
    finally:
        ll.set_precision("bf16x3")


def test_sampled_row_views_equal_the_unfused_ops():
    """sampled_row_views (gather + DropBlock(1) view + noise view + bf16 cast in two launches per class, gradient
    folded into the stacked node's d(pooled)) against index_select -> drop_pool -> noise_pool -> cat: identical
    draws, bit-identical bf16 operand, equal pooled gradient."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd.config import make_defaults
    from od_wscl_amd import precision as ll
    from od_wscl_amd.modeling.backbone.vgg16 import VGG16FC67ROIFeatureExtractor
    from od_wscl_amd.utils.device_rand import DeviceRand
    cfg = make_defaults()
    cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.POOLER_METHOD", "ROIPool", "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", 7,
                         "MODEL.ROI_BOX_HEAD.POOLER_SCALES", (0.125,), "DB.METHOD", "dropblock"])
    ll.set_precision("bf16")
    try:
        torch.manual_seed(1)
        fe = VGG16FC67ROIFeatureExtractor(cfg, 512).cuda().train()
        P = 60
        pooled0 = torch.randn(P, 512, 7, 7, device="cuda").relu_()
        rows_a = torch.tensor([3, 17, 18, 40, 59], dtype=torch.int32, device="cuda")
        rows_b = torch.tensor([0, 17, 33], dtype=torch.int32, device="cuda")          # row 17 in both classes
        groups = [(0, rows_a, 5), (0, rows_b, 3)]
        gout = torch.randn(16, 512 * 49, device="cuda")
        # fused: stacked node + row views sharing one gradient holder
        fe.rand = DeviceRand(5)
        p1 = pooled0.clone().requires_grad_(True)
        c, a = fe.forward_clean_and_aug(p1)
        x, s6, s7 = fe.sampled_row_views(p1, groups)
        n1 = fe.rand.s.next
        ((x.float() * gout).sum() + c.float().sum() + a.float().sum()).backward()
        # unfused: the torch ops on gathered rows
        fe.rand = DeviceRand(5)
        p2 = pooled0.clone().requires_grad_(True)
        c2, a2 = fe.forward_clean_and_aug(p2)
        parts, t6 = [], []
        for base, rows, k in groups:
            picked = p2.index_select(0, rows)
            drop = fe.drop_pool(picked)
            k6, k7 = fe.rand.key(), fe.rand.key()
            noisy = fe.noise_pool(picked)
            k6n, k7n = fe.rand.key(), fe.rand.key()
            parts += [drop.reshape(k, -1), noisy.reshape(k, -1)]
            t6 += [k6, k6n]
        x2 = torch.cat(parts).to(torch.bfloat16)
        assert fe.rand.s.next == n1
        assert [s[1:] for s in s6] == t6
        assert torch.equal(x.view(torch.int16), x2.view(torch.int16))
        ((torch.cat(parts) * gout.to(torch.bfloat16).float()).sum() + c2.float().sum() + a2.float().sum()).backward()
        d = (p1.grad - p2.grad).abs().max().item()
        assert d <= 2e-2 * p2.grad.abs().max().item(), d             # bf16 gradient of the stacked half on both sides
        sel = torch.tensor([3, 17, 18, 33, 40], device="cuda")
        assert (p1.grad[sel] - p2.grad[sel]).abs().max().item() <= 2e-2 * p2.grad[sel].abs().max().item()
    finally:
        ll.set_precision("bf16x3")


def test_l2norm_rows_kernels():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.nn.functional as F
    from od_wscl_amd.modeling.roi_heads.sim_head.sim_net import _L2NormRows
    x = torch.from_numpy(rng.normal(8, 1, 333 * 128).reshape(333, 128)).cuda()
    x[5] = 0                                                   # all-zero row: y = 0, gradient g / eps
    x[6] *= 1e-20
    xs = x.clone().requires_grad_(True)
    y = _L2NormRows.apply(xs, 1e-12)
    xr = x.clone().requires_grad_(True)
    ref = F.normalize(xr, dim=1)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    g = torch.from_numpy(rng.normal(9, 1, 333 * 128).reshape(333, 128)).cuda()
    g[5] = 0
    g[6] = 0
    y.backward(g)
    ref.backward(g)
    np.testing.assert_allclose(xs.grad.cpu().numpy(), xr.grad.cpu().numpy(), rtol=2e-4, atol=2e-6)


def test_pool_stack_equals_pool_then_stack(monkeypatch):
    """forward_pool_clean_and_aug (ROIPool writing the stacked bf16 operand + 16-bit argmax, sampled-row views read
    from it, all gradients scattered by one kernel) against forward_pooler -> forward_clean_and_aug -> views on the
    fp32 pooled tensor: identical draws, bit-identical operands and fc outputs, equal feature gradient."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd.config import make_defaults
    from od_wscl_amd import precision as ll
    from od_wscl_amd.modeling.backbone.vgg16 import VGG16FC67ROIFeatureExtractor
    from od_wscl_amd.structures import BoxList
    from od_wscl_amd.utils.device_rand import DeviceRand
    from od_wscl_amd import synthetic
    cfg = make_defaults()
    cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.POOLER_METHOD", "ROIPool", "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", 7,
                         "MODEL.ROI_BOX_HEAD.POOLER_SCALES", (0.125,), "DB.METHOD", "dropblock"])
    ll.set_precision("bf16")
    monkeypatch.setenv("ODW_NO_SPARSE", "1")      # this test differentiates the clean features themselves
    try:
        torch.manual_seed(2)
        fe = VGG16FC67ROIFeatureExtractor(cfg, 512).cuda().train()
        H = W = 160
        feat0 = torch.randn(2, 512, H // 8, W // 8, device="cuda").relu_().bfloat16().float()   # bf16-valued, like the backbone's
        props = []
        for k, n in enumerate((37, 29)):
            bx = torch.from_numpy(synthetic.make_proposals(3, k, n, H, W, min_size=12)).cuda()
            bx[0] = torch.tensor([150.0, 150.0, 150.4, 150.3])          # degenerate box: single-cell / empty bins
            props.append(BoxList(bx, (W, H), "xyxy"))
        rows_a = torch.tensor([1, 5, 20], dtype=torch.int32, device="cuda")
        rows_b = torch.tensor([0, 5, 28], dtype=torch.int32, device="cuda")
        groups = [(0, rows_a, 3), (37, rows_b, 3)]
        roi_index = torch.tensor([1, 5, 20, 37, 42, 65], dtype=torch.int32, device="cuda")
        gx = torch.randn(12, 512 * 49, device="cuda")
        out = {}
        for fused in (True, False):
            fe.rand = DeviceRand(11)
            f = feat0.clone().requires_grad_(True)
            if fused:
                assert fe.can_pool_stack([f])
                c, a, pooled = fe.forward_pool_clean_and_aug([f], props)
                x, s6, s7 = fe.sampled_row_views(pooled, groups, roi_index)
                stacked = pooled
            else:
                pooled = fe.forward_pooler([f], props)
                c, a = fe.forward_clean_and_aug(pooled)
                x, s6, s7 = fe.sampled_row_views(pooled, groups)
                stacked = None
            ((x.float() * gx).sum() + (c.float() ** 2).sum() + a.float().sum()).backward()
            out[fused] = (c.float(), a.float(), x.clone(), f.grad.clone(), fe.rand.s.next, s6, stacked)
        assert out[True][4] == out[False][4] and out[True][5] == out[False][5]
        assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
        assert torch.equal(out[True][2].view(torch.int16), out[False][2].view(torch.int16))
        ga, gb = out[True][3], out[False][3]
        assert (ga - gb).abs().max().item() <= 1e-4 * gb.abs().max().item() + 1e-6, (ga - gb).abs().max().item()
    finally:
        ll.set_precision("bf16x3")


def test_pool_stack_nhwc_equals_plane_form():
    """The (ROI, 64-channel) pooling forward on the NHWC bf16 map == the plane-per-workgroup form on its NCHW fp32
    copy: stacked operand (both halves) and 16-bit argmax, bit for bit (incl. empty and clipped bins)."""
    from od_wscl_amd import _lib as L, synthetic
    lib = L.lib()
    for (C, H, W, P, scale, size) in ((128, 38, 50, 300, 0.125, (304, 400)), (512, 19, 19, 200, 0.0625, (304, 304))):
        g = torch.Generator(device="cuda").manual_seed(C)
        nhwc = torch.randn(1, H, W, C, device="cuda", generator=g).bfloat16().contiguous()
        feat = nhwc.float().permute(0, 3, 1, 2).contiguous()
        bx = torch.from_numpy(synthetic.make_proposals(5, 0, P, size[0], size[1], min_size=8)).cuda()
        bx[0] = torch.tensor([-40.0, -30.0, -20.0, -10.0])           # entirely outside: every bin empty
        bx[1] = torch.tensor([0.0, 0.0, float(size[1]) + 90, float(size[0]) + 70])   # clipped at the far edges
        rois = torch.cat([torch.zeros(P, 1, device="cuda"), bx], dim=1).contiguous()
        keep = (torch.rand(P, 49, device="cuda", generator=g) > 0.25).float()
        ks = keep.sum()
        wsb = lib.odw_roi_pool_workspace(P, 7, 7)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        outs = []
        for form in ("plane", "nhwc"):
            x = torch.full((2 * P, C * 49), 7.0, dtype=torch.bfloat16, device="cuda")
            am = torch.full((P, C * 49), 5, dtype=torch.int16, device="cuda")
            if form == "plane":
                L.check(lib.odw_roi_pool_stack_forward(L.ptr(feat), L.ptr(rois), scale, 1, C, H, W, P, 7, 7, L.ptr(keep), L.ptr(ks),
                                                       L.ptr(x), x.stride(0), L.ptr(am), L.ptr(ws), wsb, L.stream()), "plane")
            else:
                wsn = lib.odw_roi_pool_stack_nhwc_workspace(P, 1, C, H, W)
                ws2 = torch.empty(wsn, dtype=torch.uint8, device="cuda")
                L.check(lib.odw_roi_pool_stack_forward_nhwc(L.ptr(nhwc), L.ptr(rois), scale, 1, C, H, W, P, L.ptr(keep), L.ptr(ks),
                                                            L.ptr(x), x.stride(0), L.ptr(am), L.ptr(ws2), wsn, L.stream()), "nhwc")
            outs.append((x, am))
        assert torch.equal(outs[0][1], outs[1][1])
        assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
        assert (outs[1][1][0] == -1).all() and (outs[1][0][0] == 0).all()


@pytest.mark.parametrize("n,h,w,bs,p", [(2000, 7, 7, 3, 0.3), (333, 7, 7, 1, 0.3), (64, 7, 7, 2, 0.5), (50, 6, 9, 5, 0.4), (1, 7, 7, 3, 0.3), (0, 7, 7, 3, 0.3)])
def test_dropblock_keep_mask_kernel_equals_the_reference_formula(n, h, w, bs, p):
    """odw_dropblock_keep_mask (one launch) against drop_block.py:38-47 evaluated with torch on the SAME uniform draw
    (the counter-based stream, od_wscl_amd/utils/rng.py on the host): centres, max-pool dilation with padding bs // 2 and
    the even-size crop, inversion, and the sum -- bit for bit."""
    import torch.nn.functional as F
    from od_wscl_amd.modeling.dropblock import DropBlock2D
    from od_wscl_amd.utils import rng
    from od_wscl_amd.utils.device_rand import DeviceRand
    db = DropBlock2D(drop_prob=p, block_size=bs)
    r = DeviceRand(77, first_stream=9, device="cuda")
    keep = db.keep_mask(n, h, w, torch.device("cuda"), r)
    assert r.s.next == 10                                                   # one logical draw, like rand.uniform(shape)
    u = torch.from_numpy(rng.uniform(77, 9, n * h * w).reshape(n, h, w)) if n else torch.zeros((0, h, w))
    centres = (u < p / (bs ** 2)).float()
    block = F.max_pool2d(centres[:, None], kernel_size=bs, stride=1, padding=bs // 2) if n else centres[:, None]
    if bs % 2 == 0 and n:
        block = block[:, :, :-1, :-1]
    want = 1 - block.squeeze(1)
    assert keep.shape == (n, h, w) and torch.equal(keep.cpu(), want)
    assert float(keep._odw_sum) == float(want.sum())
    if n:
        x = torch.randn(n, 4, h, w, device="cuda")
        got = db.train()(x, rand=DeviceRand(77, first_stream=9, device="cuda"))
        ref = x.cpu() * want[:, None] * want.numel() / want.sum()
        assert torch.allclose(got.cpu(), ref, rtol=1e-6, atol=0)


@pytest.mark.parametrize("R,C,S,ks", [(40, 64, 49, (7, 12)), (9, 512, 49, (5,)), (6, 128, 16, (3, 2))])
def test_views_written_as_planes_equal_the_fp32_views_split_afterwards(R, C, S, ks):
    """odw_rows_views_cm (bf16x2f: the sampled-row views straight as the operand of fc6) == odw_rows_drop_noise_f32 of the
    value the planes hold (x = hi + mid) followed by the split pass, bit for bit: the cell-major planes of the forward and the
    channel-major hi plane of the backward."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import _lib as L, gemm
    CS = C * S
    x = torch.from_numpy(rng.normal(31, 1, R * CS).reshape(R, CS).astype(np.float32)).cuda().relu()
    src_cm = gemm.split_rows_cm(x, C, S)                                    # (R, 2 CS): [hi | mid], k' = cell * C + channel
    hi = src_cm[:, :CS].float().view(R, S, C).permute(0, 2, 1).reshape(R, CS)
    mid = src_cm[:, CS:].float().view(R, S, C).permute(0, 2, 1).reshape(R, CS)
    x16 = (hi + mid).contiguous()                                          # what the planes hold, channel-major
    total = sum(ks)
    out_cm = torch.full((2 * total, 2 * CS), 7.0, dtype=torch.bfloat16, device="cuda")
    out_hi = torch.full((2 * total, CS), 7.0, dtype=torch.bfloat16, device="cuda")
    ref32 = torch.empty((2 * total, CS), dtype=torch.float32, device="cuda")
    sums = torch.empty(2 * len(ks), dtype=torch.float32, device="cuda")
    row0, base = 0, 0
    for gi, k in enumerate(ks):
        rows = torch.from_numpy(np.sort(np.random.RandomState(gi).choice(R - base, k, replace=False)).astype(np.int32)).cuda()
        keys = (11 + gi, 22, 33 + gi, 44)
        L.check(L.lib().odw_rows_views_cm(L.ptr(src_cm), src_cm.stride(0), CS, L.ptr(rows), base, k, C, S, 0.3, *keys,
                                          L.ptr(sums[2 * gi:]), L.ptr(out_cm), out_cm.stride(0), CS, L.ptr(out_hi), out_hi.stride(0),
                                          row0, L.stream()), "rows_views_cm")
        L.check(L.lib().odw_rows_drop_noise_f32(L.ptr(x16), L.ptr(rows), base, k, C, S, 0.3, *keys, L.ptr(sums[2 * gi + 1:]),
                                                L.ptr(ref32), ref32.stride(0), row0, L.stream()), "rows_drop_noise_f32")
        row0 += 2 * k
        base = 1
    assert torch.equal(sums[0::2], sums[1::2])
    assert torch.equal(out_hi.view(torch.int16), ref32.to(torch.bfloat16).view(torch.int16))
    assert torch.equal(out_cm.view(torch.int16), gemm.split_rows_cm(ref32, C, S).view(torch.int16))
