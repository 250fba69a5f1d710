"""Launches with device-resident extents (round 6: csrc/loss_lists.hip, od_wscl_amd/dyn.py) against the static launches of
the same extents and against the host assembly of rounds 2-5 (numpy, restated below from roi_heads/weak_head/loss.py:281-347).
-m gpu.  The bar is BIT equality wherever the two launches make the same plan: a device extent changes which workgroups run,
never what a workgroup computes."""
import ctypes

import numpy as np
import pytest
import torch

from od_wscl_amd.utils import rng

pytestmark = pytest.mark.gpu

SENT = 12345.0


@pytest.fixture(scope="module")
def mods():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import _lib, dyn, gemm, precision
    return _lib, dyn, gemm, precision


def rnd(seed, shape, scale=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy((rng.normal(seed, 1, n) * scale).reshape(shape)).cuda()


def dev_int(v):
    return torch.tensor([v], dtype=torch.int32, device="cuda")


@pytest.mark.parametrize("M,Mcap,N,K", [(880, 2048, 4096, 4096), (880, 2048, 25088, 4096), (300, 6016, 4096, 12288), (1, 512, 128, 4096),
                                        (2034, 12032, 4096, 4096), (0, 512, 128, 4096)])
def test_gemm_with_rows_on_the_device_equals_the_static_launch(mods, M, Mcap, N, K):
    L, dyn, gemm, precision = mods
    precision.set_precision("bf16")
    a = rnd(1, (Mcap, K)).bfloat16()
    b = rnd(2, (N, K), 0.05).bfloat16()
    a[M:] = float("nan")                                    # rows past the live extent may hold anything
    bias = rnd(3, (N,))
    out = torch.full((Mcap, N), SENT, device="cuda")
    m = dyn.Dyn(dev_int(M), Mcap, max(M, 1))
    dyn.gemm_nt(a, b, Mcap, N, K, out, bias=bias, relu=True, m=m)
    assert (out[M:] == SENT).all(), "rows past *m_dev were written"
    if M:
        ref = torch.empty((M, N), device="cuda")
        gemm.gemm_nt(a[:M].contiguous(), b, M, N, K, ref, bias=bias, relu=True)
        assert torch.equal(out[:M], ref)


@pytest.mark.parametrize("Kd,Kcap,M,N", [(5312, 8192, 4096, 4096), (3072, 16128, 4096, 25088), (64, 4096, 128, 4096), (2048 + 640, 6144, 4096, 4096)])
def test_gemm_with_the_reduction_length_on_the_device(mods, Kd, Kcap, M, N):
    """the weight-gradient batch: [dZ^T blocks] x [X^T blocks] over the first *k_dev columns"""
    L, dyn, gemm, precision = mods
    precision.set_precision("bf16")
    a = rnd(4, (M, Kcap), 0.1).bfloat16()
    b = rnd(5, (N, Kcap), 0.1).bfloat16()
    a[:, Kd:] = float("nan")
    b[:, Kd:] = float("nan")
    out = rnd(6, (M, N))
    ref = out.clone()
    k = dyn.Dyn(dev_int(Kd), Kcap, Kd)
    dyn.gemm_nt(a, b, M, N, Kcap, out, accumulate=True, k=k)
    gemm.gemm_nt(a[:, :Kd].contiguous(), b[:, :Kd].contiguous(), M, N, Kd, ref, accumulate=True)
    assert torch.isfinite(out).all()
    # the same plan when the hint is exact: the same bits (the static form may take its tail-column split, the dynamic one does not)
    assert (out - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()) * np.sqrt(Kd)
    # a wrong hint changes the plan, never the result beyond re-association
    out2 = ref.clone()
    out2.copy_(rnd(6, (M, N)))
    dyn.gemm_nt(a, b, M, N, Kcap, out2, accumulate=True, k=dyn.Dyn(dev_int(Kd), Kcap, Kcap))
    assert (out2 - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()) * np.sqrt(Kd)


@pytest.mark.parametrize("Kd,hint,M,N", [(448, 4096, 128, 4096), (64, 8192, 128, 4096), (1024, 8192, 4096, 4096), (192, 6144, 4096, 25088)])
def test_reduction_much_shorter_than_its_hint_leaves_empty_slices_that_are_zero(mods, Kd, hint, M, N):
    """a plan made for a long reduction (many K slices) meeting a short live length: the slices past it are EMPTY -- their
    partial tiles must be zeros, written without going through the operand pipeline (with no K step nothing waits for the
    prologue's DMA, which then lands in the LDS the epilogue stages through: tests/test_timed_step_gpu.py caught that as a
    garbage Sim_Net weight gradient, one run in three)"""
    L, dyn, gemm, precision = mods
    precision.set_precision("bf16")
    Kcap = 8192
    a = rnd(40, (M, Kcap), 0.1).bfloat16()
    b = rnd(41, (N, Kcap), 0.1).bfloat16()
    a[:, Kd:] = float("nan")
    b[:, Kd:] = float("nan")
    ref = torch.empty((M, N), device="cuda")
    gemm.gemm_nt(a[:, :Kd].contiguous(), b[:, :Kd].contiguous(), M, N, Kd, ref)
    for rep in range(6):                                     # (the failure was timing-dependent)
        out = torch.full((M, N), SENT, device="cuda")
        dyn.gemm_nt(a, b, M, N, Kcap, out, k=dyn.Dyn(dev_int(Kd), Kcap, hint))
        assert torch.isfinite(out).all()
        assert (out - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()) * np.sqrt(Kd), rep


def test_per_row_draw_table_reproduces_the_stacked_segments(mods):
    """dropout of stacked passes: the table written by odw_loss_lists_a (row inside its pass, key0, key1) draws what the
    host-side segment list drew"""
    L, dyn, gemm, precision = mods
    precision.set_precision("bf16")
    M, Mcap, N, K = 700, 1024, 4096, 512
    a = rnd(7, (Mcap, K)).bfloat16()
    b = rnd(8, (N, K), 0.05).bfloat16()
    segs = [(0, 11, 12), (200, 21, 22), (350, 31, 32), (600, 41, 42)]
    tab = np.zeros((Mcap, 4), dtype=np.uint32)
    bounds = [s[0] for s in segs] + [M]
    for i, s in enumerate(segs):
        for r in range(bounds[i], bounds[i + 1]):
            tab[r] = (r - s[0], s[1], s[2], 0)
    tab_d = torch.from_numpy(tab.view(np.int32)).cuda()
    for fn_out in (torch.float32,):
        out = torch.full((Mcap, N), SENT, device="cuda", dtype=fn_out)
        dyn.gemm_nt(a, b, Mcap, N, K, out, relu=True, drop_p=0.5, row_tab=tab_d, m=dyn.Dyn(dev_int(M), Mcap, M))
        ref = torch.empty((M, N), device="cuda", dtype=fn_out)
        gemm.gemm_nt(a[:M].contiguous(), b, M, N, K, ref, relu=True, drop_p=0.5, segs=segs)
        assert torch.equal(out[:M], ref) and (out[M:] == SENT).all()


@pytest.mark.parametrize("M,Mcap", [(446, 4032), (892, 4032), (2034, 12032), (64, 128)])
def test_cell_major_product_with_rows_on_the_device(mods, M, Mcap):
    L, dyn, gemm, precision = mods
    precision.set_precision("bf16x2f")
    C, S, N = 128, 49, 512
    K = C * S
    x = rnd(9, (Mcap, K))
    w = rnd(10, (N, K), 0.02)
    a = gemm.split_rows_cm(x, C, S)
    b = gemm.split_rows_cm(w, C, S)
    bias = rnd(11, (N,))
    tab = torch.zeros((Mcap, 4), dtype=torch.int32, device="cuda")
    tab[:, 0] = torch.arange(Mcap, dtype=torch.int32, device="cuda")
    tab[:, 1], tab[:, 2] = 77, 78
    out = torch.full((Mcap, N), SENT, device="cuda")
    dyn.gemm_nt_cm(a, b, N, C, S, out, dyn.Dyn(dev_int(M), Mcap, M), bias=bias, relu=True, drop_p=0.5, row_tab=tab)
    ref = torch.empty((M, N), device="cuda")
    gemm.gemm_nt_cm(a[:M].contiguous(), b, M, N, C, S, ref, bias=bias, relu=True, drop_p=0.5, segs=[(0, 77, 78)])
    assert torch.equal(out[:M], ref) and (out[M:] == SENT).all()


def test_prologue_and_transposes_with_rows_on_the_device(mods):
    L, dyn, gemm, precision = mods
    lib = L.lib()
    M, Mcap, N, K = 333, 1024, 4096, 512
    r64 = dyn.r64
    dy = rnd(12, (Mcap, N))
    y = torch.relu(rnd(13, (Mcap, N)))
    x = rnd(14, (Mcap, K))
    m = dyn.Dyn(dev_int(M), Mcap, M)
    # static reference
    dz_r = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    dzt_r = torch.zeros((N, r64(M)), dtype=torch.bfloat16, device="cuda")
    db_r = torch.zeros(N, device="cuda")
    L.check(lib.odw_linear_bwd_prep_part(L.ptr(dy), 3, dy.stride(0), L.ptr(y), y.stride(0), M, N, 2.0, L.ptr(dz_r), N, L.ptr(dzt_r),
                                         dzt_r.stride(0), r64(M), L.ptr(db_r), L.stream()), "prep")
    xt_r = gemm.transpose_bf16(x[:M].contiguous(), M, K)
    # dynamic, at a device column offset inside a wider matrix
    off = 448
    wide = torch.full((N, off + r64(Mcap) + 64), 7.0, dtype=torch.bfloat16, device="cuda")
    widex = torch.full((K, off + r64(Mcap) + 64), 7.0, dtype=torch.bfloat16, device="cuda")
    dz = torch.zeros((Mcap, N), dtype=torch.bfloat16, device="cuda")
    db = torch.zeros(N, device="cuda")
    col = dev_int(off)
    dyn.bwd_prep(dy, y, N, 2.0, dz, wide, db, m, tcol_off=col)
    dyn.transpose(x, K, widex, m, col_off=col)
    assert torch.equal(dz[:M], dz_r)
    assert torch.equal(wide[:, off:off + r64(M)], dzt_r) and (wide[:, :off] == 7.0).all() and (wide[:, off + r64(M):] == 7.0).all()
    assert torch.equal(widex[:, off:off + r64(M)], xt_r) and (widex[:, :off] == 7.0).all() and (widex[:, off + r64(M):] == 7.0).all()
    assert (db - db_r).abs().max().item() <= 1e-3 * db_r.abs().max().item()       # (atomic column sums: order)
    # a fused gather: rows of a larger table
    idx = torch.from_numpy(np.random.RandomState(0).permutation(Mcap)[:M].astype(np.int32)).cuda()
    idx_cap = torch.zeros(Mcap, dtype=torch.int32, device="cuda")
    idx_cap[:M] = idx
    out = torch.zeros((K, r64(Mcap)), dtype=torch.bfloat16, device="cuda")
    dyn.transpose(x, K, out, m, src_rows=idx_cap)
    assert torch.equal(out[:, :r64(M)], gemm.transpose_bf16(x[idx.long()].contiguous(), M, K))
    dz2 = torch.zeros((Mcap, N), dtype=torch.bfloat16, device="cuda")
    dzt2 = torch.zeros((N, r64(Mcap)), dtype=torch.bfloat16, device="cuda")
    dyn.bwd_prep(dy, y, N, 2.0, dz2, dzt2, None, m, y_rows=idx_cap)
    ref = (dy[:M] * (y[idx.long()] != 0) * 2.0).bfloat16()
    assert torch.equal(dz2[:M], ref)
    # bf16 input of the transpose (the views' hi plane)
    xb = x.bfloat16()
    outb = torch.zeros((K, r64(Mcap)), dtype=torch.bfloat16, device="cuda")
    dyn.transpose(xb, K, outb, m)
    assert torch.equal(outb[:, :r64(M)], gemm.transpose_bf16(xb[:M].contiguous(), M, K))


def test_split_norm_gather_scatter_with_rows_on_the_device(mods):
    L, dyn, gemm, precision = mods
    precision.set_precision("bf16x2f")
    M, Mcap, K = 517, 1024, 4096
    pa = precision.patterns("gemm")[0]
    x = rnd(15, (Mcap, K))
    m = dyn.Dyn(dev_int(M), Mcap, M)
    out = torch.full((Mcap, len(pa) * K), 3.0, dtype=torch.bfloat16, device="cuda")
    dyn.split_rows(x, pa, K, m, out=out)
    assert torch.equal(out[:M], precision.split_rows(x[:M].contiguous(), pa, K)) and (out[M:] == 3.0).all()
    e = rnd(16, (Mcap, 128))
    yn, nn = dyn.l2norm(e, m)
    ref = torch.nn.functional.normalize(e[:M], dim=1)
    assert (yn[:M] - ref).abs().max().item() < 1e-6
    g = rnd(17, (Mcap, 128))
    dx = dyn.l2norm_bwd(g, yn, nn, m)
    e2 = e[:M].clone().requires_grad_(True)
    torch.nn.functional.normalize(e2, dim=1).backward(g[:M])
    assert (dx[:M] - e2.grad).abs().max().item() < 1e-5
    # gather from two tables / scatter-add back
    t0, t1 = rnd(18, (300, 128)), rnd(19, (900, 128))
    n = 1000
    idx = torch.from_numpy(np.random.RandomState(1).randint(0, 1200, size=2048).astype(np.int32)).cuda()
    nd = dyn.Dyn(dev_int(n), 2048, n)
    got = dyn.gather_rows2(t0, t1, 300, idx, nd)
    ref = torch.cat([t0, t1])[idx[:n].long()]
    assert torch.equal(got[:n], ref)
    d0, d1 = torch.zeros_like(t0), torch.zeros_like(t1)
    gg = rnd(20, (2048, 128))
    sc = torch.tensor([0.5], device="cuda")
    dyn.scatter_rows2(gg, idx, nd, 300, d0, d1, scale=sc, alpha=0.25)
    refd = torch.zeros(1200, 128, device="cuda").index_add_(0, idx[:n].long(), gg[:n] * 0.125)
    assert (torch.cat([d0, d1]) - refd).abs().max().item() < 1e-5
    tab = rnd(21, (700, 4096)).bfloat16()
    rows = torch.from_numpy(np.random.RandomState(2).randint(0, 700, size=512).astype(np.int32)).cuda()
    got = dyn.gather_rows(tab, rows, dyn.Dyn(dev_int(400), 512, 400))
    assert torch.equal(got[:400], tab[rows[:400].long()])
    z = torch.ones((512, 128), device="cuda")
    dyn.zero_rows(z, dyn.Dyn(dev_int(100), 512, 100))
    assert (z[:100] == 0).all() and (z[100:] == 1).all()


@pytest.mark.parametrize("N", [96, 857, 1500, 2793, 3755])
def test_supcon_with_n_on_the_device_equals_the_static_launch(mods, N):
    L, dyn, gemm, precision = mods
    from od_wscl_amd import _C
    cap = 4096
    F_ = torch.nn.functional.normalize(rnd(22, (cap, 128)), dim=1).contiguous()
    y = torch.from_numpy((np.arange(cap) % 5).astype(np.int32)).cuda()
    w = torch.from_numpy(rng.uniform(23, 1, cap)).cuda()
    loss, dF = dyn.supcon(F_, y, w, 0.2, dyn.Dyn(dev_int(N), cap, N))
    rl, rd = _C.supcon_v2(F_[:N].contiguous(), y[:N].contiguous(), w[:N].contiguous(), 0.2)
    assert torch.equal(loss.reshape(()), rl.reshape(())) and torch.equal(dF[:N], rd)


def _views_inputs(P_, C, S, seed):
    x = rnd(seed, (P_, C * S))
    from od_wscl_amd import gemm
    return gemm.split_rows_cm(x, C, S)


def test_grouped_views_equal_the_per_group_launches(mods):
    """the drop / noise views of every (image, class) group in ONE launch (group sizes on the device) = the per-group launches
    with host-side sizes, forward and backward (fc_extractor._RowViews)"""
    L, dyn, gemm, precision = mods
    lib = L.lib()
    P_, C, S = 600, 128, 49
    CS = C * S
    src = _views_inputs(P_, C, S, 31)
    ks = [37, 5, 120]
    rs = np.random.RandomState(3)
    rows_h = [np.sort(rs.choice(P_, size=k, replace=False)).astype(np.int32) for k in ks]
    base = [0, 0, 0]
    keys = [(101 + i, 102 + i, 201 + i, 202 + i) for i in range(3)]
    E = sum(ks)
    Ecap = 512
    # per-group reference
    ref_cm = torch.zeros((2 * E, 2 * CS), dtype=torch.bfloat16, device="cuda")
    ref_hi = torch.zeros((2 * E, CS), dtype=torch.bfloat16, device="cuda")
    sums_r = torch.zeros(3, device="cuda")
    row0 = 0
    for g in range(3):
        rd = torch.from_numpy(rows_h[g]).cuda()
        L.check(lib.odw_rows_views_cm(L.ptr(src), src.stride(0), CS, L.ptr(rd), base[g], ks[g], C, S, 0.3, keys[g][0], keys[g][1],
                                      keys[g][2], keys[g][3], L.ptr(sums_r[g:]), L.ptr(ref_cm), ref_cm.stride(0), CS, L.ptr(ref_hi),
                                      ref_hi.stride(0), row0, L.stream()), "views")
        row0 += 2 * ks[g]
    # grouped
    e0 = torch.tensor(np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)).cuda()
    src_row = torch.zeros(Ecap, dtype=torch.int32, device="cuda")
    src_row[:E] = torch.from_numpy(np.concatenate(rows_h)).cuda()
    keys_d = torch.from_numpy(np.asarray(keys, dtype=np.uint32).view(np.int32).reshape(-1)).cuda()
    n_e = dev_int(E)
    out_cm = torch.full((2 * Ecap, 2 * CS), 9.0, dtype=torch.bfloat16, device="cuda")
    out_hi = torch.full((2 * Ecap, CS), 9.0, dtype=torch.bfloat16, device="cuda")
    sums = torch.zeros(3, device="cuda")
    L.check(lib.odw_rows_views_cm_grouped(L.ptr(src), src.stride(0), CS, 3, Ecap, L.ptr(n_e), L.ptr(e0), L.ptr(keys_d), L.ptr(src_row),
                                          C, S, 0.3, L.ptr(sums), L.ptr(out_cm), out_cm.stride(0), CS, L.ptr(out_hi), out_hi.stride(0),
                                          L.stream()), "views grouped")
    assert torch.equal(sums, sums_r)
    assert torch.equal(out_cm[:2 * E], ref_cm) and torch.equal(out_hi[:2 * E], ref_hi)
    assert (out_cm[2 * E:] == 9.0).all() and (out_hi[2 * E:] == 9.0).all()
    # backward into the side buffer, entries stored behind a device offset
    dx = rnd(32, (2 * Ecap, CS))
    iota = torch.arange(Ecap, dtype=torch.int32, device="cuda")
    ext_r = torch.zeros((E, CS), device="cuda")
    row0, e_at = 0, 0
    for g in range(3):
        L.check(lib.odw_rows_drop_noise_bwd_store(L.ptr(dx), 1, dx.stride(0), row0, L.ptr(iota), e_at, ks[g], C, S, 0.3, keys[g][0],
                                                  keys[g][1], keys[g][2], keys[g][3], L.ptr(sums_r[g:]), L.ptr(ext_r), L.stream()), "bwd")
        row0 += 2 * ks[g]
        e_at += ks[g]
    A = 70
    ext = torch.full((A + Ecap, CS), 5.0, device="cuda")
    a_d = dev_int(A)
    L.check(lib.odw_rows_views_bwd_store_grouped(L.ptr(dx), 1, dx.stride(0), 3, Ecap, L.ptr(n_e), L.ptr(e0), L.ptr(keys_d), L.ptr(sums),
                                                 C, S, 0.3, L.ptr(a_d), L.ptr(ext), L.stream()), "bwd grouped")
    assert torch.equal(ext[A:A + E], ext_r) and (ext[:A] == 5.0).all() and (ext[A + E:] == 5.0).all()


# ---- the host assembly of rounds 2-5 (loss_fused.py), restated with numpy: the checker of csrc/loss_lists.hip --------------
def host_lists(sizes, pos_host, counts_h, rows_h, fresh_cnt_h, fresh_rows_h, final_score, colsum, C):
    n_img = len(sizes)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    sum_p = int(offs[-1])
    meta, row0 = [], 0
    for idx in range(n_img):
        for ci, c in enumerate(pos_host[idx]):
            k = int(counts_h[idx][ci])
            meta.append((idx, ci, c, k, row0, rows_h[idx][ci][:k].astype(np.int64)))
            row0 += 2 * k
    classes = sorted(set(c for pc in pos_host for c in pc))
    bank_parts = {c: [] for c in classes}
    for (idx, ci, c, k, r0, r_h) in meta:
        bank_parts[c] += [r_h + offs[idx], np.arange(sum_p + r0, sum_p + r0 + 2 * k)]
    bank_index = {c: np.concatenate(bank_parts[c]) for c in classes}
    bank_off, bank_cnt, pos = [0] * (C - 1), [0] * (C - 1), 0
    for c in classes:
        bank_off[c], bank_cnt[c] = pos, len(bank_index[c])
        pos += len(bank_index[c])
    roi_index = np.concatenate([m[5] + offs[m[0]] for m in meta])
    feat, labels = [], []
    for c in classes:
        ix = [bank_index[c]]
        for idx in range(n_img):
            if c in pos_host[idx]:
                ci = pos_host[idx].index(c)
                for i in range(3):
                    ix.append(fresh_rows_h[idx][i][ci][:fresh_cnt_h[idx][i][ci]].astype(np.int64) + offs[idx])
        ix = np.concatenate(ix)
        feat.append(ix)
        labels.append(np.full(len(ix), c))
    feat_all = np.concatenate(feat)
    w = []
    for (idx, ci, c, k, r0, r_h) in meta:
        v = final_score[r_h + offs[idx], c + 1] / colsum[idx][c + 1]
        w += [v, v, v]
    for idx in range(n_img):
        for i in range(3):
            for ci, c in enumerate(pos_host[idx]):
                f = fresh_rows_h[idx][i][ci][:fresh_cnt_h[idx][i][ci]].astype(np.int64) + offs[idx]
                w.append(final_score[f, c + 1] / colsum[idx][c + 1])
    is_prop = feat_all < sum_p
    act = np.unique(feat_all[is_prop])
    return dict(meta=meta, bank_all=np.concatenate([bank_index[c] for c in classes]), bank_off=bank_off, bank_cnt=bank_cnt,
                roi_index=roi_index, feat_all=feat_all, labels=np.concatenate(labels), weights=np.concatenate(w).astype(np.float32),
                act=act, is_prop=is_prop, sum_p=sum_p)


@pytest.mark.parametrize("case", [([300], [[4]]), ([300], [[2, 7, 11]]), ([260, 300, 200], [[3, 9], [9], [1, 3, 9]])])
def test_device_lists_equal_the_host_assembly(mods, case):
    """odw_loss_lists_a / _b on synthetic selection results against the numpy assembly of loss_fused.py (rounds 2-5), list by
    list: entry prefix, ROI list, class banks, per-row draw tables, SupCon features / labels / weights (Q1 order), the
    re-attached rows, the side buffer's entry list and every derived scalar"""
    L, dyn, gemm, precision = mods
    lib = L.lib()
    sizes, pos_host = case
    n_img, C = len(sizes), 21
    max_p, sum_p = max(sizes), sum(sizes)
    maxpos = max(len(p) for p in pos_host)
    rs = np.random.RandomState(7)
    counts_h = np.zeros((n_img, maxpos), dtype=np.int32)
    rows_h = np.zeros((n_img, maxpos, max_p), dtype=np.int32)
    fresh_cnt_h = np.zeros((n_img, 3, maxpos), dtype=np.int32)
    fresh_rows_h = np.zeros((n_img, 3, maxpos, max_p), dtype=np.int32)
    for idx in range(n_img):
        for ci in range(len(pos_host[idx])):
            k = int(rs.randint(1, 60))
            counts_h[idx, ci] = k
            rows_h[idx, ci, :k] = np.sort(rs.choice(sizes[idx], size=k, replace=False))
            for i in range(3):
                n = int(rs.randint(1, 9))
                fresh_cnt_h[idx, i, ci] = n
                fresh_rows_h[idx, i, ci, :n] = np.sort(rs.choice(sizes[idx], size=n, replace=False))
    final_score = rs.rand(sum_p, C).astype(np.float32) + 0.1
    colstat = (rs.rand(n_img, 3, 128).astype(np.float32) + 0.5)
    ref = host_lists(sizes, pos_host, counts_h, rows_h, fresh_cnt_h, fresh_rows_h, final_score, colstat[:, 2, :], C)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    groups = [(idx, ci, c) for idx in range(n_img) for ci, c in enumerate(pos_host[idx])]
    G = len(groups)
    grp = np.zeros((G, 16), dtype=np.int64)
    for g, (idx, ci, c) in enumerate(groups):
        grp[g, :12] = (idx, ci, c, offs[idx], 1000 + g, 2000 + g, 3000 + g, 4000 + g, 5000 + g, 6000 + g, 7000 + g, 8000 + g)
    order = sorted(range(G), key=lambda g: (groups[g][2], g))
    cu = lambda a, dt=np.int32: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).cuda()
    grp_d, order_d = cu(grp.reshape(-1)), cu(order)
    E_cap = 64 * ((G * max_p + 63) // 64)
    scal_a = torch.zeros(16, dtype=torch.int32, device="cuda")
    e0 = torch.zeros(G + 1, dtype=torch.int32, device="cuda")
    roi_index = torch.zeros(E_cap, dtype=torch.int32, device="cuda")
    bank_index = torch.zeros(3 * E_cap, dtype=torch.int32, device="cuda")
    bank_off = torch.zeros(C - 1, dtype=torch.int32, device="cuda")
    bank_cnt = torch.zeros(C - 1, dtype=torch.int32, device="cuda")
    tabs = torch.zeros((2, 2 * E_cap, 4), dtype=torch.int32, device="cuda")
    counts_d, rows_d, fresh_rows_d, fresh_cnt_d, offs_d = cu(counts_h), cu(rows_h), cu(fresh_rows_h), cu(fresh_cnt_h), cu(offs)
    L.check(lib.odw_loss_lists_a(L.ptr(grp_d), L.ptr(order_d), G, L.ptr(counts_d), L.ptr(rows_d), maxpos, max_p, sum_p, C - 1,
                                 E_cap, L.ptr(scal_a), L.ptr(e0), L.ptr(roi_index), L.ptr(bank_index), L.ptr(bank_off), L.ptr(bank_cnt),
                                 L.ptr(tabs[0]), L.ptr(tabs[1]), L.stream()), "lists_a")
    sa = scal_a.cpu().numpy()
    E1 = sum(m[3] for m in ref["meta"])
    assert list(sa[:5]) == [E1, 2 * E1, (2 * E1 + 63) // 64 * 64, 3 * E1, 0]
    assert list(e0.cpu().numpy()) == [0] + list(np.cumsum([m[3] for m in ref["meta"]]))
    assert np.array_equal(roi_index.cpu().numpy()[:E1], ref["roi_index"])
    assert np.array_equal(bank_index.cpu().numpy()[:3 * E1], ref["bank_all"])
    assert list(bank_off.cpu().numpy()) == ref["bank_off"] and list(bank_cnt.cpu().numpy()) == ref["bank_cnt"]
    t6, t7 = tabs[0].cpu().numpy().view(np.uint32), tabs[1].cpu().numpy().view(np.uint32)
    for g, m in enumerate(ref["meta"]):
        k, r0 = m[3], m[4]
        assert np.array_equal(t6[r0:r0 + 2 * k, 0], np.concatenate([np.arange(k), np.arange(k)]))
        assert (t6[r0:r0 + k, 1] == 1000 + g).all() and (t6[r0:r0 + k, 2] == 2000 + g).all()
        assert (t7[r0:r0 + k, 1] == 3000 + g).all() and (t7[r0 + k:r0 + 2 * k, 1] == 7000 + g).all()
        assert (t6[r0 + k:r0 + 2 * k, 1] == 5000 + g).all() and (t6[r0 + k:r0 + 2 * k, 2] == 6000 + g).all()
    # ---- lists B
    N = len(ref["feat_all"])
    A = len(ref["act"])
    N_cap, A_cap = N + 100, 64 * ((sum_p + 63) // 64)
    p64 = 64 * ((sum_p + 63) // 64)
    scal_b = torch.zeros(16, dtype=torch.int32, device="cuda")
    feat_index = torch.zeros(N_cap, dtype=torch.int32, device="cuda")
    labels = torch.zeros(N_cap, dtype=torch.int32, device="cuda")
    weights = torch.zeros(N_cap, dtype=torch.float32, device="cuda")
    act_rows = torch.zeros(A_cap, dtype=torch.int32, device="cuda")
    roi_all = torch.zeros(A_cap + E_cap, dtype=torch.int32, device="cuda")
    n_pos = cu([len(p) for p in pos_host])
    pos_cls = cu([p + [0] * (maxpos - len(p)) for p in pos_host])
    gt_cnt = torch.zeros((n_img, 3), dtype=torch.int32, device="cuda")
    sticky = torch.zeros(2, dtype=torch.int32, device="cuda")
    fs_d, cs_d = torch.from_numpy(final_score).cuda(), torch.from_numpy(colstat).cuda()
    L.check(lib.odw_loss_lists_b(L.ptr(grp_d), L.ptr(order_d), G, L.ptr(offs_d), L.ptr(n_pos), L.ptr(pos_cls), n_img, maxpos, max_p,
                                 sum_p, L.ptr(scal_a), L.ptr(e0), L.ptr(roi_index), L.ptr(bank_index), L.ptr(bank_off), L.ptr(bank_cnt),
                                 L.ptr(fresh_rows_d), L.ptr(fresh_cnt_d), L.ptr(gt_cnt), 2048, L.ptr(fs_d), C, L.ptr(cs_d),
                                 3 * 128, 2 * 128, N_cap, A_cap, E_cap, p64, L.ptr(scal_b), L.ptr(feat_index), L.ptr(labels),
                                 L.ptr(weights), L.ptr(act_rows), L.ptr(roi_all), L.ptr(sticky), L.stream()), "lists_b")
    assert sticky.cpu().tolist() == [0, 0]
    sb = scal_b.cpu().numpy()
    r64 = lambda v: (v + 63) // 64 * 64
    V = 2 * E1
    assert list(sb[:11]) == [N, A, A + E1, 0, p64 + r64(V), p64 + r64(V) + r64(A), r64(V), r64(V) + r64(A), r64(A), 0, r64(N)]
    assert np.array_equal(act_rows.cpu().numpy()[:A], ref["act"])
    assert np.array_equal(roi_all.cpu().numpy()[:A + E1], np.concatenate([ref["act"], ref["roi_index"]]))
    remap = np.where(ref["is_prop"], np.searchsorted(ref["act"], np.minimum(ref["feat_all"], sum_p - 1)), ref["feat_all"] - sum_p + A_cap)
    assert np.array_equal(feat_index.cpu().numpy()[:N], remap)
    assert np.array_equal(labels.cpu().numpy()[:N], ref["labels"])
    assert np.array_equal(weights.cpu().numpy()[:N], ref["weights"])        # the same IEEE division
    # overflow is flagged, never written past the capacity
    small = 16
    scal_o = torch.zeros(16, dtype=torch.int32, device="cuda")
    roi_o = torch.full((small + 8,), -5, dtype=torch.int32, device="cuda")
    bank_o = torch.zeros(3 * small + 8, dtype=torch.int32, device="cuda")
    e0_o, boff_o, bcnt_o, tabs_o = e0.clone(), bank_off.clone(), bank_cnt.clone(), tabs.clone()
    L.check(lib.odw_loss_lists_a(L.ptr(grp_d), L.ptr(order_d), G, L.ptr(counts_d), L.ptr(rows_d), maxpos, max_p, sum_p, C - 1,
                                 small, L.ptr(scal_o), L.ptr(e0_o), L.ptr(roi_o), L.ptr(bank_o), L.ptr(boff_o),
                                 L.ptr(bcnt_o), L.ptr(tabs_o[0]), L.ptr(tabs_o[1]), L.stream()), "lists_a small")
    so = scal_o.cpu().numpy()
    assert (so[4] == 1 and so[0] == small) if E1 > small else so[4] == 0
    assert (roi_o[small:] == -5).all()


def test_pooling_backward_with_the_entry_count_on_the_device(mods):
    L, dyn, gemm, precision = mods
    lib = L.lib()
    B, C, H, W, R, ph, pw = 1, 64, 24, 24, 50, 7, 7
    K = C * ph * pw
    from od_wscl_amd import synthetic
    boxes = synthetic.make_proposals(5, 0, R, 192, 192, min_size=12)
    rois = torch.from_numpy(np.concatenate([np.zeros((R, 1), np.float32), boxes], 1)).cuda()
    argmax = torch.from_numpy(np.random.RandomState(4).randint(0, H * W, size=(R, K)).astype(np.int16)).cuda()
    keep = torch.ones((R, ph * pw), device="cuda")
    ksum = keep.sum().reshape(1)
    dx = rnd(41, (2 * R, K))
    E, Ecap = 23, 64
    extra = rnd(42, (Ecap, K))
    extra[E:] = float("nan")
    eroi = torch.from_numpy(np.random.RandomState(5).randint(0, R, size=Ecap).astype(np.int32)).cuda()
    ws = torch.zeros(64, dtype=torch.uint8, device="cuda")
    g_ref = torch.empty((B, C, H, W), device="cuda")
    extra_live = extra[:E].contiguous()
    L.check(lib.odw_roi_pool_stack_backward_ws(L.ptr(dx), 1, K, L.ptr(argmax), L.ptr(rois), L.ptr(keep), L.ptr(ksum), L.ptr(extra_live),
                                               L.ptr(eroi), E, 1, B, C, H, W, R, ph, pw, L.ptr(g_ref), L.ptr(ws), 64, L.stream()), "bwd")
    g = torch.empty((B, C, H, W), device="cuda")
    e_d = dev_int(E)
    L.check(lib.odw_roi_pool_stack_backward_dyn(L.ptr(dx), 1, K, L.ptr(argmax), L.ptr(rois), L.ptr(keep), L.ptr(ksum), L.ptr(extra),
                                                L.ptr(eroi), Ecap, L.ptr(e_d), 1, B, C, H, W, R, ph, pw, L.ptr(g), L.ptr(ws), 64,
                                                L.stream()), "bwd dyn")
    assert torch.equal(g, g_ref)


def _absbits(t):
    return int((t.float().contiguous().view(torch.int32) & 0x7fffffff).max().item())


@pytest.mark.parametrize("M,N,K", [(2000, 25088, 4096),       # fc6's input gradient: 256 x 256 tiles + the tail columns' split-K product
                                   (300, 1024, 8192),          # split-K: the reduction pass takes the maximum
                                   (100, 132, 72),             # the register-staged kernel's element stores
                                   (513, 4096, 256), (64, 64, 64)])
def test_input_gradient_product_leaves_the_maximum_the_prepass_would_find(mods, M, N, K):
    """odw_gemm_nt_bf16_absmax: the same C as the plain launch, bit for bit, and the bit pattern of max |C| in the word -- what
    odwfx::absmax_kernel finds by re-reading C (a maximum does not depend on the order it is taken in)."""
    L, dyn, gemm, precision = mods
    precision.set_precision("bf16")
    a = rnd(61, (M, K), 0.3).bfloat16()
    b = rnd(62, (N, K), 0.05).bfloat16()
    ref = torch.full((M, N), SENT, device="cuda")
    gemm.gemm_nt(a, b, M, N, K, ref, alpha=0.5)
    out = torch.full((M, N), SENT, device="cuda")
    word = torch.zeros(16, dtype=torch.int32, device="cuda")
    gemm.gemm_nt(a, b, M, N, K, out, alpha=0.5, absmax=word)
    assert torch.equal(out, ref)
    assert int(word[0].item()) == _absbits(ref), (hex(int(word[0].item())), hex(_absbits(ref)))
    assert int(word[1:].abs().max().item()) == 0
    # the word only grows: a second product onto a word that already holds more leaves it alone
    big = torch.full((16,), 0x7f000000, dtype=torch.int32, device="cuda")
    gemm.gemm_nt(a, b, M, N, K, out, alpha=0.5, absmax=big)
    assert int(big[0].item()) == 0x7f000000


@pytest.mark.parametrize("skip_clean,with_extra", [(1, True), (0, True), (1, False)])
def test_pooling_backward_with_the_maximum_already_taken(mods, skip_clean, with_extra):
    """odw_roi_pool_stack_backward_scaled (max |dX| left in the workspace word by the GEMM that wrote dX; only the side buffer
    is scanned) = odw_roi_pool_stack_backward_ws / _dyn, bit for bit."""
    L, dyn, gemm, precision = mods
    lib = L.lib()
    B, C, H, W, R, ph, pw = 1, 64, 24, 24, 50, 7, 7
    K = C * ph * pw
    from od_wscl_amd import synthetic
    boxes = synthetic.make_proposals(5, 0, R, 192, 192, min_size=12)
    rois = torch.from_numpy(np.concatenate([np.zeros((R, 1), np.float32), boxes], 1)).cuda()
    argmax = torch.from_numpy(np.random.RandomState(4).randint(0, H * W, size=(R, K)).astype(np.int16)).cuda()
    keep = torch.ones((R, ph * pw), device="cuda")
    ksum = keep.sum().reshape(1)
    dx = rnd(41, (2 * R, K))
    if skip_clean:
        dx[:R] = float("nan")                   # the clean half is unset in the sparse backward: never read
    E, Ecap = (23, 64) if with_extra else (0, 0)
    extra = rnd(42, (Ecap, K), 3.0) if with_extra else None         # (larger than dX: the side buffer's scan must still land)
    eroi = torch.from_numpy(np.random.RandomState(5).randint(0, R, size=max(Ecap, 1)).astype(np.int32)).cuda() if with_extra else None
    ws = torch.zeros(64, dtype=torch.uint8, device="cuda")
    g_ref = torch.empty((B, C, H, W), device="cuda")
    extra_live = extra[:E].contiguous() if with_extra else None
    L.check(lib.odw_roi_pool_stack_backward_ws(L.ptr(dx), 1, K, L.ptr(argmax), L.ptr(rois), L.ptr(keep), L.ptr(ksum), L.ptr(extra_live),
                                               L.ptr(eroi), E, skip_clean, B, C, H, W, R, ph, pw, L.ptr(g_ref), L.ptr(ws), 64,
                                               L.stream()), "bwd")
    live = dx[R:] if skip_clean else dx
    word = torch.zeros(16, dtype=torch.int32, device="cuda")
    word[0] = _absbits(live)
    g = torch.empty((B, C, H, W), device="cuda")
    L.check(lib.odw_roi_pool_stack_backward_scaled(L.ptr(dx), 1, K, L.ptr(argmax), L.ptr(rois), L.ptr(keep), L.ptr(ksum),
                                                   L.ptr(extra_live), L.ptr(eroi), E, None, skip_clean, B, C, H, W, R, ph, pw,
                                                   L.ptr(g), L.ptr(word), 64, L.stream()), "bwd scaled")
    assert torch.equal(g, g_ref)
    if with_extra:                              # the device-resident entry count
        extra[E:] = float("nan")
        word.zero_()
        word[0] = _absbits(live)
        g2 = torch.empty((B, C, H, W), device="cuda")
        L.check(lib.odw_roi_pool_stack_backward_scaled(L.ptr(dx), 1, K, L.ptr(argmax), L.ptr(rois), L.ptr(keep), L.ptr(ksum),
                                                       L.ptr(extra), L.ptr(eroi), Ecap, L.ptr(dev_int(E)), skip_clean, B, C, H, W, R,
                                                       ph, pw, L.ptr(g2), L.ptr(word), 64, L.stream()), "bwd scaled dyn")
        assert torch.equal(g2, g_ref)
