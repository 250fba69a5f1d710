"""The shared clean + DropBlock forward of the first head Linear (csrc/gemm_bf16.hip: gemm_nt_cm_kernel; csrc/split.hip:
split_rows_cm_kernel).  ROIWeakRegHead.forward evaluates fc6 on the pooled features and on their DropBlock view
(roi_heads/weak_head/weak_head.py:107-112, modeling/dropblock/drop_block.py:38-50); the kernel produces both from one sweep
over the clean operand.  References: numpy for the plane layout (bit-exact), torch fp64 for the products, and the stacked
evaluation the kernel replaces (same dropout draws: the zero pattern must agree)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf16_rn(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


@pytest.fixture(autouse=True)
def _mode():
    from od_wscl_amd import precision
    precision.set_precision("bf16x2f")
    yield
    precision.set_precision("bf16")


@pytest.mark.parametrize("R,C,S", [(5, 64, 49), (33, 128, 9), (7, 512, 49), (3, 64, 64), (2, 64, 1)])
def test_split_rows_cm_is_the_permuted_plane_decomposition(R, C, S):
    from od_wscl_amd import gemm
    rs = np.random.RandomState(R * 100 + S)
    x = (rs.randn(R, C * S) * np.exp(rs.randn(R, C * S) * 3)).astype(np.float32)
    x[0, :3] = [0.0, np.inf, -0.0]
    hi = _bf16_rn(x)
    with np.errstate(invalid="ignore"):
        mid = _bf16_rn(np.where(np.isfinite(x), x - hi, 0.0).astype(np.float32))
    out = gemm.split_rows_cm(torch.from_numpy(x).cuda(), C, S).float().cpu().numpy()
    K = C * S
    perm = lambda a: a.reshape(R, C, S).transpose(0, 2, 1).reshape(R, K)
    np.testing.assert_array_equal(out[:, :K], perm(hi))
    np.testing.assert_array_equal(out[:, K:], perm(mid))


def _operands(seed, M, N, C, S):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.relu(torch.randn(M, C * S, device="cuda", generator=g)) * 0.7
    w = torch.randn(N, C * S, device="cuda", generator=g) * 0.02
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    return x, w, b


@pytest.mark.parametrize("M,N,C,S", [(130, 200, 64, 9), (257, 357, 128, 49), (1000, 128, 64, 4), (40, 4096, 512, 49),
                                     (600, 256, 64, 49)])
def test_cm_product_matches_fp64_with_and_without_the_cell_split(M, N, C, S):
    from od_wscl_amd import gemm, _lib as L
    x, w, b = _operands(M + N, M, N, C, S)
    xc, wc = gemm.split_rows_cm(x, C, S), gemm.split_rows_cm(w, C, S)
    ref = torch.relu(x.double() @ w.double().T + b.double())
    tol = 3e-5 * max(1.0, ref.abs().max().item())
    out = torch.full((M, N), float("nan"), device="cuda")
    gemm.gemm_nt_cm(xc, wc, M, N, C, S, out, bias=b, relu=True)                    # (workspace: small M splits over cells)
    assert (out.double() - ref).abs().max().item() <= tol
    one = torch.full((M, N), float("nan"), device="cuda")
    K = C * S
    L.check(L.lib().odw_gemm_nt_cm(L.ptr(xc), 2 * K, K, L.ptr(wc), 2 * K, K, M, N, C, S, None, None, 0, L.ptr(one), N, L.ptr(b), 1,
                                   0.0, 0, None, None, None, None, 0, L.stream()), "one pass")
    assert (one.double() - ref).abs().max().item() <= tol
    assert (one - out).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("P,N,C,S", [(100, 256, 64, 49), (300, 357, 128, 9), (513, 128, 64, 49), (1500, 512, 64, 49)])
def test_pair_form_is_the_stacked_clean_and_dropblock_evaluation(P, N, C, S):
    """rows [0, P): fc6 of the clean features; rows [P, 2P): fc6 of x * keep * numel / sum -- against the stacked product
    over the reference-order planes (gemm_nt: what rounds 1-3 ran) with the same dropout keys, and against fp64."""
    from od_wscl_amd import gemm, precision
    x, w, b = _operands(P + N, P, N, C, S)
    K = C * S
    g = torch.Generator(device="cuda").manual_seed(P)
    keep = (torch.rand(P, S, device="cuda", generator=g) > 0.45).float()
    keep[0] = 1.0                                   # nothing dropped
    keep[1] = 0.0                                   # every cell dropped: bias only
    keep[2] = 0.0
    keep[2, S - 1] = 1.0                            # only the last cell kept
    ksum = keep.sum()
    xd = (x.view(P, C, S) * keep[:, None, :] * keep.numel() / ksum).reshape(P, K)
    pa, pb = precision.patterns("gemm")
    segs = [(0, 11, 12), (P, 13, 14)]
    ref = torch.empty(2 * P, N, device="cuda")
    gemm.gemm_nt(precision.split_rows(torch.cat([x, xd]), pa, K), precision.split_rows(w, pb, K), 2 * P, N, 3 * K, ref, bias=b,
                 relu=True, drop_p=0.5, segs=segs)
    out = torch.full((2 * P, N), float("nan"), device="cuda")
    gemm.gemm_nt_cm(gemm.split_rows_cm(x, C, S), gemm.split_rows_cm(w, C, S), P, N, C, S, out, bias=b, relu=True, drop_p=0.5,
                    segs=segs, keep=keep, keep_sum=ksum, drop_row0=P)
    assert not torch.isnan(out).any()
    y64 = torch.relu(torch.cat([x, xd]).double() @ w.double().T + b.double())
    scale = max(1.0, y64.abs().max().item())
    # the same dropout draws: an entry is zero in one exactly when it is zero in the other, up to ReLU ties at ~0
    differ = (out == 0) != (ref == 0)
    assert differ.float().mean().item() <= 1e-5
    assert ((out - ref).abs() * (~differ)).max().item() <= 1e-4 * scale
    kept = out != 0
    assert ((out.double() - 2.0 * y64).abs() * kept).max().item() <= 6e-5 * scale          # (dropout keeps x 1 / (1 - 0.5))
    assert (out[P + 1][out[P + 1] != 0] - 2.0 * torch.relu(b)[out[P + 1] != 0]).abs().max().item() <= 1e-6      # bias only


@pytest.mark.parametrize("P,N,C,S", [(100, 256, 64, 49), (300, 4096, 128, 49)])
def test_pair_form_split_over_cells_equals_the_single_sweep(P, N, C, S):
    """Few ROIs: the pair form splits the cells over blockIdx.y (partial clean / DropBlock sums + one reduction pass).  Same
    results as the unsplit launch up to the re-association of the fp32 sums; identical dropout pattern."""
    from od_wscl_amd import gemm, _lib as L
    x, w, b = _operands(P * 3 + N, P, N, C, S)
    K = C * S
    g = torch.Generator(device="cuda").manual_seed(P + 1)
    keep = (torch.rand(P, S, device="cuda", generator=g) > 0.45).float()
    ksum = keep.sum()
    xc, wc = gemm.split_rows_cm(x, C, S), gemm.split_rows_cm(w, C, S)
    segs = [(0, 21, 22), (P, 23, 24)]
    assert L.lib().odw_gemm_nt_cm_pair_workspace(P, N, S) > 0
    split = torch.full((2 * P, N), float("nan"), device="cuda")
    gemm.gemm_nt_cm(xc, wc, P, N, C, S, split, bias=b, relu=True, drop_p=0.5, segs=segs, keep=keep, keep_sum=ksum, drop_row0=P)
    one = torch.full((2 * P, N), float("nan"), device="cuda")
    rows = (ctypes.c_int * 4)(0, P, 0, 0)
    keys = (ctypes.c_uint32 * 8)(21, 22, 23, 24, 0, 0, 0, 0)
    L.check(L.lib().odw_gemm_nt_cm(L.ptr(xc), 2 * K, K, L.ptr(wc), 2 * K, K, P, N, C, S, L.ptr(keep), L.ptr(ksum), P, L.ptr(one), N,
                                   L.ptr(b), 1, 0.5, 2, ctypes.cast(rows, ctypes.c_void_p), ctypes.cast(keys, ctypes.c_void_p), None,
                                   None, 0, L.stream()), "single sweep")
    assert not torch.isnan(split).any() and not torch.isnan(one).any()
    differ = (split == 0) != (one == 0)
    assert differ.float().mean().item() <= 1e-5
    assert ((split - one).abs() * (~differ)).max().item() <= 3e-5 * max(1.0, one.abs().max().item())


def test_c_abi_binding_of_integration_5a_gives_the_references_two_fc6_evaluations():
    """INTEGRATION.md 5a, as a maintainer of the reference would call it: raw C-ABI entry points on (P, 512, 7, 7) pooled
    features and DropBlock2D's block mask, against `relu(fc6(x))` and `relu(fc6(x * mask * mask.numel() / mask.sum()))`
    (vgg16.py:121, drop_block.py:49-50) evaluated by torch in fp64."""
    from od_wscl_amd import _lib as L
    lib = L.lib()
    P, C, S, N = 96, 512, 49, 4096
    g = torch.Generator(device="cuda").manual_seed(5)
    pooled = torch.relu(torch.randn(P, C, 7, 7, device="cuda", generator=g))
    weight = torch.randn(N, C * S, device="cuda", generator=g) * 0.01
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    keep = (torch.rand(P, 7, 7, device="cuda", generator=g) > 0.4).float()
    x_cm = torch.empty((P, 2 * C * S), dtype=torch.bfloat16, device="cuda")
    w_cm = torch.empty((N, 2 * C * S), dtype=torch.bfloat16, device="cuda")
    L.check(lib.odw_split_rows_cm(L.ptr(pooled.reshape(P, -1)), C * S, P, C, S, L.ptr(x_cm), 2 * C * S, C * S, L.stream()), "x")
    L.check(lib.odw_split_rows_cm(L.ptr(weight), C * S, N, C, S, L.ptr(w_cm), 2 * C * S, C * S, L.stream()), "w")
    out = torch.full((2 * P, N), float("nan"), device="cuda")
    ksum = keep.sum()
    L.check(lib.odw_gemm_nt_cm(L.ptr(x_cm), 2 * C * S, C * S, L.ptr(w_cm), 2 * C * S, C * S, P, N, C, S, L.ptr(keep), L.ptr(ksum), P,
                               L.ptr(out), N, L.ptr(bias), 1, 0.0, 0, None, None, None, None, 0, L.stream()), "pair")
    x64, w64, b64 = pooled.double(), weight.double(), bias.double()
    clean = torch.relu(x64.reshape(P, -1) @ w64.T + b64)
    xd = x64 * keep.double()[:, None] * keep.numel() / keep.double().sum()
    drop = torch.relu(xd.reshape(P, -1) @ w64.T + b64)
    scale = max(1.0, clean.abs().max().item(), drop.abs().max().item())
    assert (out[:P].double() - clean).abs().max().item() <= 3e-5 * scale
    assert (out[P:].double() - drop).abs().max().item() <= 3e-5 * scale


def test_pair_form_refuses_what_it_cannot_do():
    from od_wscl_amd import gemm
    x, w, b = _operands(1, 64, 128, 64, 9)
    xc, wc = gemm.split_rows_cm(x, 64, 9), gemm.split_rows_cm(w, 64, 9)
    out = torch.empty(128, 128, device="cuda")
    keep = torch.ones(64, 9, device="cuda")
    with pytest.raises(RuntimeError):               # the DropBlock half must not overlap the clean rows
        gemm.gemm_nt_cm(xc, wc, 64, 128, 64, 9, out, keep=keep, keep_sum=keep.sum(), drop_row0=32)
    with pytest.raises(RuntimeError):               # channel count not a multiple of the K tile
        gemm.gemm_nt_cm(xc, wc, 64, 128, 48, 12, out)
