"""Arithmetic precision of the MFMA products (Linear layers and convolutions) of the hot path.

  "bf16"   : operands rounded to bf16 once, fp32 accumulation -- the throughput mode (BASELINE.json north_star:
             "backbone convs and the ROI-head FC use MFMA bf16 tiles").
  "bf16x3" : fp32-grade.  Every operand is carried as three bf16 planes hi + mid + lo laid out along the reduction
             axis and the SAME MFMA kernels accumulate the six plane products of order <= 2 (csrc/split.hip) --
             the reference's arithmetic (config/defaults.py:559 DTYPE float32) to ~2^-24 per product, on the bf16
             matrix cores (2.5 PF / 6 = 417 TF-equivalent against 157 TF for the fp32-input MFMA).  Activations
             stay fp32 between kernels.  This is the mode in which the 1e-3 loss / bit-exact selection parity with
             the reference's goldens is asserted (tests/test_e2e_gpu.py).
  "bf16x2" : two planes, three products (hi.hi + hi.mid + mid.hi): ~2^-16 per product at half the cost of bf16x3.
  "bf16x2f": the bench headline since round 3.  FORWARD products (convolutions, fc6/fc7, Sim_Net, predictor, the
             sampled-row views) as in "bf16x2" -- losses within 1e-3 of the reference and every index selection
             identical (tests/test_e2e_gpu.py, tests/test_fullsize_gpu.py assert BASELINE.json's bar in this mode) --
             BACKWARD products (input and weight gradients) on single bf16 planes: gradients are not part of that
             bar, and the backward is half of the step's MFMA work.  Activations and their gradients are fp32 tensors
             between kernels; the backward kernels round their operands to bf16 as they read them.

One process-wide setting: every kernel of a step must agree on the activation dtype between kernels."""
import ctypes

import torch

from . import _lib as L

_MODE = "bf16"

# plane codes along the reduction axis: 0 = hi, 1 = mid, 2 = lo, 3 = zeros.  The t-th block of operand A meets the
# t-th block of operand B, so (A[t], B[t]) enumerates the plane products that are summed.
_PATTERNS = {
    # GEMM operands: any number of blocks
    ("bf16x3", "gemm"): ((0, 0, 0, 1, 1, 2), (0, 1, 2, 0, 1, 0)),
    ("bf16x2", "gemm"): ((0, 0, 1), (0, 1, 0)),
    # convolution operands: the implicit-GEMM kernel wants a power-of-two channel count -> pad with zero blocks
    ("bf16x3", "conv"): ((0, 0, 0, 1, 1, 2, 3, 3), (0, 1, 2, 0, 1, 0, 3, 3)),
    ("bf16x2", "conv"): ((0, 0, 1, 1), (0, 1, 0, 1)),
    ("bf16x2f", "gemm"): ((0, 0, 1), (0, 1, 0)),
    ("bf16x2f", "conv"): ((0, 0, 1, 1), (0, 1, 0, 1)),
}

MODES = ("bf16", "bf16x3", "bf16x2", "bf16x2f")


def set_precision(name):
    global _MODE
    if name not in MODES:
        raise ValueError("precision %r (%s)" % (name, " | ".join(MODES)))
    _MODE = name


def get_precision():
    return _MODE


def split_mode():
    """True when the FORWARD operands are split into bf16 planes (activations are fp32 between kernels)."""
    return _MODE != "bf16"


def bwd_split():
    """True when the BACKWARD products run on split planes too ("bf16x3", "bf16x2"); False in "bf16" and in
    "bf16x2f", whose backward rounds each operand to one bf16 plane."""
    return _MODE in ("bf16x3", "bf16x2")


def act_dtype():
    return torch.float32 if split_mode() else torch.bfloat16


def patterns(kind="gemm"):
    """(pattern of operand A, pattern of operand B) for the current mode."""
    return _PATTERNS[(_MODE, kind)]


def conv_patterns(cp):
    """(pattern of the activation operand, pattern of the packed weight) of a 3x3 convolution over `cp` (padded) input
    channels.  "bf16x2f" runs the three useful plane products when three blocks of cp channels make a multiple of 64
    (the halo-tile kernel's granule) -- the four-block form carries a fourth, mid.mid, only because the 128x128
    kernel decodes (tap, channel) with shifts and wants a power of two."""
    if _MODE == "bf16x2f" and (3 * cp) % 64 == 0:
        return _PATTERNS[("bf16x2f", "gemm")]
    return _PATTERNS[(_MODE, "conv")]


def r64(n):
    return (n + 63) // 64 * 64


_PAT_C = {}


def _c_pattern(pat):
    a = _PAT_C.get(pat)
    if a is None:
        a = _PAT_C[pat] = (ctypes.c_int * len(pat))(*pat)
    return a


def split_rows(x, pat, block=None, out=None):
    """x (R, C) fp32 -> (R, T*block) bf16: block t holds plane pat[t] of x, zero padded from C to `block`
    (default: C rounded up to 64).  Reduction along the columns of x."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    R, C = x.shape
    block = r64(C) if block is None else block
    T = len(pat)
    if out is None:
        out = torch.empty((R, T * block), dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().odw_split_rows_bf16(L.ptr(x), x.stride(0), R, C, ctypes.cast(_c_pattern(pat), ctypes.c_void_p), T,
                                        L.ptr(out), out.stride(0), block, L.stream()), "split_rows_bf16")
    return out


def split_cols(x, pat, block=None, out=None):
    """x (R, C) fp32 -> (C, T*block) bf16: out[c][t*block + r] = plane pat[t] of x[r][c], zero padded from R to
    `block` (default: R rounded up to 64).  Reduction along the rows of x.  `out` may be a column-block view of a
    wider matrix (weight-gradient batches)."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    R, C = x.shape
    block = r64(R) if block is None else block
    T = len(pat)
    if out is None:
        out = torch.empty((C, T * block), dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().odw_split_cols_bf16(L.ptr(x), x.stride(0), R, C, ctypes.cast(_c_pattern(pat), ctypes.c_void_p), T,
                                        L.ptr(out), out.stride(0), block, L.stream()), "split_cols_bf16")
    return out


def pack_conv_weight(rows, pat, block, n_out):
    """rows (n_out*9, C) fp32 = a 3x3 weight laid out (output, tap, channel) -> the implicit-GEMM operand
    (n_out, r64(9 * T * block)) bf16: per tap the T plane blocks of the channel axis, rows zero padded to a multiple
    of 64 (the kernel reads whole 64-element K tiles)."""
    w = split_rows(rows, pat, block).view(n_out, -1)
    k = w.shape[1]
    if k % 64:
        w = torch.nn.functional.pad(w, (0, r64(k) - k))
    return w.contiguous()


def bwd_mask(dy, y, scale, db=None, out=None):
    """dZ = dY * [Y != 0] * scale (Y None: dZ = dY) in fp32 and db += column sums of dZ (deterministic two-stage
    reduction); `out` may alias dy.  dy (M, N) fp32 with unit column stride."""
    M, N = dy.shape
    dz = torch.empty((M, N), dtype=torch.float32, device=dy.device) if out is None else out
    ws_bytes = L.lib().odw_linear_bwd_mask_workspace(M, N) if db is not None else 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device) if ws_bytes else None
    L.check(L.lib().odw_linear_bwd_mask_f32(L.ptr(dy), dy.stride(0), L.ptr(y), 0, y.stride(0) if y is not None else 0, M, N,
                                            float(scale), L.ptr(dz), dz.stride(0), L.ptr(db), L.ptr(ws), ws_bytes, L.stream()),
            "linear_bwd_mask_f32")
    return dz
