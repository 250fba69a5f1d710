"""Launches whose extents live on the device (round 6): Python front end of the `_dyn` / `_grouped` entry points of
include/odwscl.h.

The OD-WSCL loss picks its rows from the step's scores (roi_heads/weak_head/loss.py:281-347): how many sampled rows,
how many discoveries, how many clean rows the contrastive loss differentiates is only known on the GPU.  Rounds 2-5
read those counts back (two blocking host reads per step) to shape every tensor exactly; here a tensor is allocated
for the CAPACITY of its extent, the live value stays in device memory (`Dyn.t`, a one-element int32 view written by
csrc/loss_lists.hip) and every kernel reads it when it starts.  The host never waits for the GPU.

`Dyn.hint` is what the extent is expected to be -- the value of an earlier step, read back without blocking.  It picks
kernel variants and split-K factors, nothing else: any hint is correct."""
import ctypes

import torch

from . import _lib as L
from . import precision as P
from .utils.kernel_timer import kernel_timer

_VARIANT_SYMBOL = {0: "", 1: "glds_", 2: "ring_", 3: "big_"}


def r64(n):
    return (n + 63) // 64 * 64


class Dyn(object):
    """A device-resident extent: t = int32 tensor of ONE element holding the live value; cap = the most it can be (what
    exists as memory, what launches are sized for); hint = what it is expected to be."""
    __slots__ = ("t", "cap", "hint")

    def __init__(self, t, cap, hint=None):
        assert t.dtype == torch.int32 and t.numel() == 1
        self.t, self.cap = t, int(cap)
        self.hint = int(min(max(1, hint if hint else cap), cap))


_PLAN = {}
_ARENA = {}


def workspace(nbytes, device):
    """Split-K partials of the dynamic launches: ONE arena per device, grown geometrically and never shrunk.  A plan made
    for a new hint may want a larger workspace than any step before; from the caching allocator that is a hipMalloc in the
    middle of a step, from the arena it is a view.  Launches on one stream use it one after the other."""
    if not nbytes:
        return None
    key = (str(device), L.stream().value)       # (per STREAM: two streams' launches run at the same time)
    a = _ARENA.get(key)
    if a is None or a.numel() < nbytes:
        grow = max(int(nbytes * 1.5), 256 << 20)
        if a is not None:
            torch.cuda.synchronize(device)      # (a launch on another stream may still read the old arena: growth is rare)
        a = None
        _ARENA.pop(key, None)
        a = _ARENA[key] = torch.empty((grow + (1 << 20) - 1) >> 20 << 20, dtype=torch.uint8, device=device)
    return a[:nbytes]


def gemm_nt(a, b, M_cap, N, K_cap, out, bias=None, relu=False, alpha=1.0, drop_p=0.0, row_tab=None, m=None, k=None,
            accumulate=False, planes=1, tag=None):
    """out[:*m, :N] (+)= epilogue(alpha a[:*m, :*k] b[:N, :*k]^T) (gemm.gemm_nt with device-resident M and / or K)."""
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
    assert (m is not None or k is not None) and (drop_p == 0.0 or row_tab is not None)
    out_bf16 = out.dtype == torch.bfloat16
    mh, kh = (m.hint if m is not None else M_cap), (k.hint if k is not None else K_cap)
    lib = L.lib()
    key = (M_cap, mh, m is None, N, K_cap, kh, a.stride(0), b.stride(0), out.stride(0), out_bf16, out.data_ptr() & 15)
    plan = _PLAN.get(key)
    if plan is None:
        var = ctypes.c_int(0)
        ws_bytes = lib.odw_gemm_nt_bf16_dyn_workspace(M_cap, mh if m is not None else 0, N, K_cap, kh, a.stride(0), b.stride(0), L.ptr(out), out.stride(0),
                                                      1 if out_bf16 else 0, ctypes.byref(var))
        if len(_PLAN) > 4096:
            _PLAN.clear()
        plan = _PLAN[key] = (ws_bytes, var.value)
    ws_bytes, var = plan
    ws = workspace(ws_bytes, a.device)
    split = ws is not None
    sym = "gemm_nt_bf16_%skernel<%s>%s" % (_VARIANT_SYMBOL[var], ("false" if split or not out_bf16 else "true")
                                           + (", 7" if var == 3 else ""), " split-K+reduce" if split else "")
    kernel_timer.layer = tag
    with kernel_timer.region(sym, flops=2.0 * mh * N * kh, alg=2.0 * mh * N * kh / planes, shape="M~%d,N=%d,K~%d (device extents)" % (mh, N, kh)):
        L.check(lib.odw_gemm_nt_bf16_dyn(L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), M_cap, N, K_cap, L.ptr(out), out.stride(0),
                                         1 if out_bf16 else 0, L.ptr(bias), 1 if relu else 0, float(alpha), float(drop_p),
                                         L.ptr(row_tab), L.ptr(m.t) if m is not None else None, mh,
                                         L.ptr(k.t) if k is not None else None, kh, 1 if accumulate else 0, L.ptr(ws), ws_bytes,
                                         L.stream()), "gemm_nt_bf16_dyn")
    kernel_timer.layer = None
    return out


def gemm_nt_cm(a_cm, b_cm, N, C, S, out, m, bias=None, relu=False, drop_p=0.0, row_tab=None, tag=None):
    """The plain product over cell-major planes (gemm.gemm_nt_cm without keep) over the first *m rows of a_cm."""
    K = C * S
    M_cap = m.cap
    assert a_cm.shape[1] == 2 * K and b_cm.shape[1] == 2 * K and out.dtype == torch.float32 and a_cm.shape[0] >= M_cap
    lib = L.lib()
    ws_bytes = lib.odw_gemm_nt_cm_dyn_workspace(M_cap, m.hint, N, S)
    ws = workspace(ws_bytes, out.device)
    sym = "gemm_nt_cm_kernel<false, 1>%s" % (" split+reduce" if ws_bytes else "")
    kernel_timer.layer = tag
    with kernel_timer.region(sym, flops=2.0 * m.hint * N * 3 * K, alg=2.0 * m.hint * N * K,
                             shape="M~%d,N=%d,C=%d,S=%d (device extent)" % (m.hint, N, C, S)):
        L.check(lib.odw_gemm_nt_cm_dyn(L.ptr(a_cm), a_cm.stride(0), K, L.ptr(b_cm), b_cm.stride(0), K, M_cap, N, C, S, L.ptr(out),
                                       out.stride(0), L.ptr(bias), 1 if relu else 0, float(drop_p), L.ptr(row_tab), L.ptr(m.t),
                                       m.hint, L.ptr(ws), ws_bytes, L.stream()), "gemm_nt_cm_dyn")
    kernel_timer.layer = None
    return out


def split_rows(x, pat, block, m, out=None):
    """precision.split_rows over the first *m rows."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    R, C = x.shape
    T = len(pat)
    if out is None:
        out = torch.empty((R, T * block), dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().odw_split_rows_bf16_dyn(L.ptr(x), x.stride(0), min(R, m.cap), C, ctypes.cast(P._c_pattern(pat), ctypes.c_void_p), T,
                                            L.ptr(out), out.stride(0), block, L.ptr(m.t), L.stream()), "split_rows_bf16_dyn")
    return out


def transpose(x, cols, out, m, col_off=None, src_rows=None, rows_cap=None):
    """out[c][col_off + r] = bf16(x[src_rows[r] if src_rows else r][c]), r < *m, zero padded to r64(*m); out = the whole
    (cols x ld) matrix (col_off = a device int tensor, or None = 0)."""
    rows_cap = m.cap if rows_cap is None else rows_cap
    L.check(L.lib().odw_transpose_to_bf16_dyn(L.ptr(x), 1 if x.dtype == torch.float32 else 0, x.stride(0), rows_cap, cols, L.ptr(out),
                                              out.stride(0), L.ptr(m.t), L.ptr(col_off), L.ptr(src_rows), L.stream()),
            "transpose_to_bf16_dyn")
    return out


def bwd_prep(dy, y, N, scale, dz, dzt, db, m, tcol_off=None, y_rows=None):
    """gemm._backward_single_plane's prologue over *m rows: dz (m.cap x ld_z) row-major, dzt = the whole (N x ld_t) matrix the
    transposed block lands in at column *tcol_off; y_rows: the mask source row of row r is y[y_rows[r]]."""
    flags = (1 if dy.dtype == torch.float32 else 0) | (2 if (y is not None and y.dtype == torch.float32) else 0)
    L.check(L.lib().odw_linear_bwd_prep_dyn(L.ptr(dy), flags, dy.stride(0), L.ptr(y), y.stride(0) if y is not None else 0, m.cap, N,
                                            float(scale), L.ptr(dz), dz.stride(0), L.ptr(dzt), dzt.stride(0), L.ptr(db), L.ptr(m.t),
                                            L.ptr(tcol_off), L.ptr(y_rows), L.stream()), "linear_bwd_prep_dyn")


def l2norm(x, m, eps=1e-12):
    y = torch.empty_like(x)
    norm = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    L.check(L.lib().odw_l2norm_rows_dyn(L.ptr(x), min(x.shape[0], m.cap), x.shape[1], eps, L.ptr(y), L.ptr(norm), L.ptr(m.t), L.stream()),
            "l2norm_rows_dyn")
    return y, norm


def l2norm_bwd(g, y, norm, m, eps=1e-12):
    dx = torch.empty_like(g)
    L.check(L.lib().odw_l2norm_rows_bwd_dyn(L.ptr(g), L.ptr(y), L.ptr(norm), min(y.shape[0], m.cap), y.shape[1], eps, L.ptr(dx),
                                            L.ptr(m.t), L.stream()), "l2norm_rows_bwd_dyn")
    return dx


def gather_rows2(t0, t1, split, index, n, out=None):
    D = t0.shape[1]
    if out is None:
        out = torch.empty((n.cap, D), dtype=torch.float32, device=t0.device)
    L.check(L.lib().odw_gather_rows2_dyn(L.ptr(t0), L.ptr(t1), int(split), L.ptr(index), L.ptr(n.t), n.cap, D, L.ptr(out), L.stream()),
            "gather_rows2_dyn")
    return out


def scatter_rows2(g, index, n, split, d0, d1, scale=None, alpha=1.0):
    L.check(L.lib().odw_scatter_rows2_dyn(L.ptr(g), L.ptr(index), L.ptr(n.t), n.cap, g.shape[1], int(split), L.ptr(scale), float(alpha),
                                          L.ptr(d0), L.ptr(d1), L.stream()), "scatter_rows2_dyn")


def gather_rows(src, index, n, out=None):
    """out[r] = src[index[r]] for r < *n (rows of any dtype whose byte width is a multiple of 16)."""
    rb = src.shape[1] * src.element_size()
    if out is None:
        out = torch.empty((n.cap, src.shape[1]), dtype=src.dtype, device=src.device)
    L.check(L.lib().odw_gather_rows_dyn(L.ptr(src), src.stride(0) * src.element_size(), L.ptr(index), L.ptr(n.t), n.cap, rb,
                                        L.ptr(out), out.stride(0) * out.element_size(), L.stream()), "gather_rows_dyn")
    return out


def zero_rows(t, n):
    L.check(L.lib().odw_zero_rows_dyn(L.ptr(t), t.stride(0) * t.element_size(), t.shape[1] * t.element_size(), L.ptr(n.t),
                                      min(n.cap, t.shape[0]), L.stream()), "zero_rows_dyn")
    return t


def supcon(F, labels, weights, temperature, n, grad_scale=1.0):
    """_C.supcon_v2 over the first *n rows -> (loss (1,), dF (n.cap, D))."""
    N_cap, D = F.shape
    loss = torch.empty((1,), dtype=torch.float32, device=F.device)
    dF = torch.empty_like(F)
    nbytes = L.lib().odw_supcon_dyn_workspace(N_cap)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=F.device)
    L.check(L.lib().odw_supcon_v2_dyn(L.ptr(F), L.ptr(labels), L.ptr(weights), N_cap, D, float(temperature), float(grad_scale),
                                      L.ptr(loss), L.ptr(dF), L.ptr(n.t), L.ptr(ws), nbytes, L.stream()), "supcon_v2_dyn")
    return loss, dF
