"""Python front-end of the bf16 MFMA GEMM (csrc/gemm_bf16.hip) and the fused Linear autograd op.

A Linear layer keeps its fp32 master weight (the parameter the optimiser and checkpoints see,
reference names unchanged) plus two bf16 shadows the matrix cores read: W (N x K) for the
forward product and W^T (K x N8) for the input gradient.  Shadows are refreshed when the
parameter's version counter moves (after every optimiser step).
"""
import ctypes

import torch

from . import _lib as L
from . import precision as P
from .utils.kernel_timer import kernel_timer
from .utils.step_trace import step_trace


MAX_SEGS = 8      # dropout row segments one GEMM launch carries keys for (kMaxSeg in csrc/gemm_bf16.hip)


def _r8(n):
    return (n + 7) // 8 * 8


def _r64(n):
    """Leading dimension of scratch operands: the LDS-DMA GEMM reads whole 64-element K tiles, so the
    reduction dimension is zero padded to a multiple of 64."""
    return (n + 63) // 64 * 64


_VARIANT_SYMBOL = {0: "", 1: "glds_", 2: "ring_", 3: "big_"}      # names as rocprofv3's kernel trace prints them
_PLAN_CACHE = {}      # (M, N, K, lda, ldb, ldc, out_bf16, alignment of C) -> (split-K workspace bytes, kernel variant)


def gemm_nt(a, b, M, N, K, out, bias=None, relu=False, alpha=1.0, drop_p=0.0, segs=None, accumulate=False, row_ids=None,
            planes=1, absmax=None):
    """out[M,N] (+)= epilogue(alpha * a[M,K] @ b[N,K]^T); a, b bf16 2-D tensors (row stride % 8 == 0).
    planes: K spans this many bf16 plane products of ONE fp32-grade product (split precision modes) -- only the timer's
    bookkeeping uses it (algorithmic FLOPs = issued / planes).
    absmax: a zeroed int32 device word -- the plain product (fp32 out, N % 4 == 0) also leaves the bit pattern of max |out|
    there (odw_gemm_nt_bf16_absmax: the consumer that scatters `out` in fixed point no longer re-reads it for its scale)."""
    L.need_gpu(a, b, out)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    assert a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
    nseg = len(segs) if segs else 0
    assert nseg <= MAX_SEGS, "at most %d stacked passes per GEMM launch" % MAX_SEGS
    rows = (ctypes.c_int * MAX_SEGS)(*([s[0] for s in segs] + [0] * (MAX_SEGS - nseg))) if nseg else None
    keys = (ctypes.c_uint32 * (2 * MAX_SEGS))(*([k for s in segs for k in (s[1], s[2])] + [0] * (2 * (MAX_SEGS - nseg)))) if nseg else None
    out_bf16 = out.dtype == torch.bfloat16
    # per-symbol timing for bench.py's roofline object (same names as rocprofv3's kernel trace)
    # the planner's answer depends on the shape, the strides and the alignment of C only: asked once per distinct product
    # (a ctypes round trip per launch otherwise, ~70 launches per step)
    out_ptr = L.ptr(out)
    # (ODW_GEMM_VARIANT / ODW_GEMM_SPLITK changed mid-process: the launch itself re-plans in C; a stale entry here only means
    # an unused or missing split-K workspace and a stale timing label)
    key = (M, N, K, a.stride(0), b.stride(0), out.stride(0), out_bf16, out.data_ptr() & 15)
    plan = _PLAN_CACHE.get(key)
    if plan is None:
        step_trace.count("gemm_plan_miss")
        var = ctypes.c_int(0)
        ws_bytes = L.lib().odw_gemm_nt_bf16_workspace(M, N, K, a.stride(0), b.stride(0), out_ptr, out.stride(0),
                                                      1 if out_bf16 else 0, ctypes.byref(var))
        if len(_PLAN_CACHE) > 4096:
            _PLAN_CACHE.clear()
        plan = _PLAN_CACHE[key] = (ws_bytes, var.value)
    ws_bytes, var = plan
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device) if ws_bytes else None     # split-K partials
    split = ws is not None
    sym = "gemm_nt_bf16_%skernel<%s>%s" % (_VARIANT_SYMBOL[var], ("false" if split or not out_bf16 else "true")
                                           + (", 7" if var == 3 else ""), " split-K+reduce" if split else "")
    if absmax is not None:
        assert not out_bf16 and bias is None and not relu and drop_p == 0.0 and not accumulate and row_ids is None and N % 4 == 0
        with kernel_timer.region(sym, flops=2.0 * M * N * K, alg=2.0 * M * N * K / planes, shape="M=%d,N=%d,K=%d" % (M, N, K)):
            L.check(L.lib().odw_gemm_nt_bf16_absmax(L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), M, N, K, L.ptr(out),
                                                    out.stride(0), float(alpha), L.ptr(absmax), L.ptr(ws), ws_bytes, L.stream()),
                    "gemm_nt_bf16_absmax")
        return out
    with kernel_timer.region(sym, flops=2.0 * M * N * K, alg=2.0 * M * N * K / planes, shape="M=%d,N=%d,K=%d" % (M, N, K)):
        L.check(L.lib().odw_gemm_nt_bf16_ws(L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), M, N, K, L.ptr(out),
                                            out.stride(0), 1 if out_bf16 else 0, L.ptr(bias), 1 if relu else 0,
                                            float(alpha), float(drop_p), nseg,
                                            ctypes.cast(rows, ctypes.c_void_p) if nseg else None,
                                            ctypes.cast(keys, ctypes.c_void_p) if nseg else None, L.ptr(row_ids),
                                            1 if accumulate else 0, L.ptr(ws), ws_bytes, L.stream()), "gemm_nt_bf16")
    return out


def to_bf16(x):
    """fp32 -> bf16 copy (round to nearest even) on the device."""
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().odw_f32_to_bf16(L.ptr(x), L.ptr(out), x.numel(), L.stream()), "f32_to_bf16")
    return out


def transpose_bf16(x, rows, cols, out=None):
    """(rows x cols) fp32|bf16 -> (cols x r64(rows)) bf16, zero padded (into `out` when given)."""
    ld = _r64(rows)
    if out is None:
        out = torch.empty((cols, ld), dtype=torch.bfloat16, device=x.device)
    assert out.shape == (cols, ld) and out.dtype == torch.bfloat16
    L.check(L.lib().odw_transpose_to_bf16(L.ptr(x), 1 if x.dtype == torch.float32 else 0, x.stride(0), rows, cols,
                                          L.ptr(out), ld, L.stream()), "transpose_to_bf16")
    return out


def split_rows_cm(x, C, S, out=None):
    """x (R, C*S) fp32, k = c*S + s (the reference's flattening of (C, 7, 7)) -> (R, 2*C*S) bf16: the two cell-major planes
    [hi | mid], k' = s*C + c (csrc/split.hip: split_rows_cm_kernel)."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[1] == C * S
    R, K = x.shape
    if out is None:
        out = torch.empty((R, 2 * K), dtype=torch.bfloat16, device=x.device)
    assert out.shape == (R, 2 * K) and out.dtype == torch.bfloat16 and out.stride(1) == 1
    L.check(L.lib().odw_split_rows_cm(L.ptr(x), x.stride(0), R, C, S, L.ptr(out), out.stride(0), K, L.stream()), "split_rows_cm")
    return out


def gemm_nt_cm(a_cm, b_cm, M, N, C, S, out, bias=None, relu=False, drop_p=0.0, segs=None, row_ids=None, keep=None,
               keep_sum=None, drop_row0=0):
    """out = epilogue(A B^T) over cell-major planes [hi | mid] (three plane products, csrc/gemm_bf16.hip: gemm_nt_cm_kernel).
    keep (M x S) given: the PAIR form -- rows [0, M) the clean product, rows [drop_row0, drop_row0 + M) the product for
    x * keep * (M S / keep_sum), from one sweep."""
    L.need_gpu(a_cm, b_cm, out)
    K = C * S
    assert a_cm.dtype == torch.bfloat16 and b_cm.dtype == torch.bfloat16 and out.dtype == torch.float32
    assert a_cm.shape[1] == 2 * K and b_cm.shape[1] == 2 * K and a_cm.stride(1) == 1 and b_cm.stride(1) == 1 and out.stride(1) == 1
    nseg = len(segs) if segs else 0
    assert nseg <= MAX_SEGS
    rows = (ctypes.c_int * MAX_SEGS)(*([s[0] for s in segs] + [0] * (MAX_SEGS - nseg))) if nseg else None
    keys = (ctypes.c_uint32 * (2 * MAX_SEGS))(*([k for s in segs for k in (s[1], s[2])] + [0] * (2 * (MAX_SEGS - nseg)))) if nseg else None
    pair = keep is not None
    ws_bytes = (L.lib().odw_gemm_nt_cm_pair_workspace(M, N, S) if drop_row0 == M else 0) if pair else L.lib().odw_gemm_nt_cm_workspace(M, N, S)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=out.device) if ws_bytes else None
    split = bool(ws_bytes) and L.lib().odw_gemm_nt_cm_workspace(M, N, S) > 0
    sym = "gemm_nt_cm_kernel<%s, 1>%s" % ("true" if pair else "false", " split+reduce" if split else "")      # rocprofv3's name
    # flops = MFMA work ISSUED (three plane products of one sweep); alg = the reference's arithmetic for what the launch
    # delivers: one fp32 fc6 product, or -- pair form -- two (the clean and the DropBlock evaluation, weak_head.py:107-112)
    with kernel_timer.region(sym, flops=2.0 * M * N * 3 * K, alg=2.0 * M * N * K * (2 if pair else 1),
                             shape="M=%d,N=%d,C=%d,S=%d%s" % (M, N, C, S, ",pair" if pair else "")):
        L.check(L.lib().odw_gemm_nt_cm(L.ptr(a_cm), a_cm.stride(0), K, L.ptr(b_cm), b_cm.stride(0), K, M, N, C, S,
                                       L.ptr(keep), L.ptr(keep_sum), drop_row0, L.ptr(out), out.stride(0), L.ptr(bias),
                                       1 if relu else 0, float(drop_p), nseg,
                                       ctypes.cast(rows, ctypes.c_void_p) if nseg else None,
                                       ctypes.cast(keys, ctypes.c_void_p) if nseg else None, L.ptr(row_ids),
                                       L.ptr(ws), ws_bytes, L.stream()), "gemm_nt_cm")
    return out


class Shadow(object):
    """bf16 copies of one fp32 weight the matrix cores read: w (N x K) for the forward product and wt (K x r64(N))
    for the input gradient.  In a split precision mode (precision.py) the same matrices as bf16 planes along the
    reduction axis: w (N x T*r64(K)), wt (K x T*r64(N)); in "bf16x2f" only the forward operand w is split, wt is the
    single-plane transposed copy the bf16 backward reads.

    A shadow the optimiser manages (engine.FlatSGD keeps w / wt up to date itself after every step) is bound to
    the precision mode it was built in: using it under another mode raises instead of reading a stale layout."""

    def __init__(self, weight):
        self.weight = weight
        self.version = -1
        self.mode = None
        self.w = None
        self.wt = None
        self.managed = False      # True: the optimiser refreshes w / wt itself (engine.FlatSGD)
        self.batch = None         # gemm.WgradBatch: one weight-gradient GEMM per step over all evaluations
        self.bwd2 = False         # "bf16x2f" with ODW_BWD2=1 (a measurement mode): THIS layer's backward products run on two
                                  # planes per operand as well (wt = planes of W^T); see bwd2_layer()
        self.cm = None            # (C, S): the weight is ALSO kept as cell-major planes w_cm (N x 2K: [hi | mid], k' = s*C + c)
        self.w_cm = None          # for the shared clean + DropBlock forward of the first head Linear (pair_linear)

    def build(self, w_out=None, wt_out=None):
        """(Re)build w / wt from the fp32 master in the current mode (into the given buffers when they fit)."""
        w = self.weight
        n, k = w.shape
        assert k % 8 == 0, "in_features must be a multiple of 8"
        wd = w.detach()
        wd = wd if wd.stride(1) == 1 else wd.contiguous()
        with torch.no_grad():
            if self.cm is not None and P.get_precision() == "bf16x2f":
                # the first head Linear in "bf16x2f": every forward reads the cell-major planes (gemm_nt_cm), no other copy
                self.w_cm = split_rows_cm(wd, self.cm[0], self.cm[1], out=self.w_cm)
                self.w = None
                self.wt = transpose_bf16(wd, n, k, out=wt_out)
            elif P.split_mode():
                pb = P.patterns("gemm")[1]
                self.w = P.split_rows(wd, pb, _r64(k), out=w_out)
                self.wt = P.split_cols(wd, pb, _r64(n), out=wt_out) if (P.bwd_split() or self.bwd2) else transpose_bf16(wd, n, k, out=wt_out)
                self.w_cm = None
            else:
                self.w = to_bf16(wd)
                self.wt = transpose_bf16(wd, n, k, out=wt_out)
                self.w_cm = None
        self.version = w._version
        self.mode = P.get_precision()

    # engine.FlatSGD: the event behind which the optimiser's side stream has rewritten this weight's copies; the first
    # evaluation that reads them waits for it (instead of the whole next forward waiting at the end of the step)
    pending = None

    def refresh(self):
        if self.pending is not None:
            torch.cuda.current_stream().wait_event(self.pending)
            self.pending = None
        mode = P.get_precision()
        if self.managed:
            if self.mode != mode:
                raise RuntimeError("gemm.Shadow: this weight's copies are kept by an optimiser built in precision mode %r; "
                                   "the process-wide mode is now %r (build a new training step after set_precision)"
                                   % (self.mode, mode))
            return self
        if self.mode == mode and self.version == self.weight._version and (self.w is not None or self.w_cm is not None):
            return self
        self.build()
        return self


def bwd2_layer(weight):
    """ODW_BWD2=1 (measurement): in "bf16x2f" the backward products of the SMALL-K Linears -- fc7, Sim_Net,
    the predictor: in_features <= 4096 -- run on two bf16 planes per operand (three plane products) like the forward,
    instead of one.  fc6 (K = 25088) and the convolutions keep the single-plane backward.  profiles/r05/bwd2_report.txt
    holds the step time and the per-tensor gradient error with and without."""
    import os
    return os.environ.get("ODW_BWD2") == "1" and P.get_precision() == "bf16x2f" and weight.dim() == 2 and weight.shape[1] <= 4096


# Set to a list by a caller that wants the input-gradient GEMM of ONE layer (`deferred_weight`: the layer whose input
# gradient nothing in the backward it is about to run consumes) handed back as a closure instead of launched (see
# _backward_single_plane.input_gradient); None = launch everything in place.
deferred_dgrad = None
deferred_weight = None


class WgradBatch(object):
    """One weight-gradient GEMM per step for a Linear that is evaluated several times (fc6: the stacked pass, the
    sampled-row views, the re-evaluated clean rows).  dW = sum_e dZ_e^T X_e = [dZ_1; dZ_2; ...]^T [X_1; X_2; ...]:
    every evaluation's backward lays its dZ^T and X^T as a column block of two shared matrices and the LAST one
    runs the GEMM over all rows -- instead of one read-modify-write pass over the (411 MB for fc6) gradient per
    evaluation.  Evaluations register at forward time; engine.FlatSGD flushes a step that ended early."""

    # columns kept free behind the blocks registered when the buffers are first asked for: an evaluation may register
    # AFTER another one's backward has run (the fused loss runs the dense losses' backward while the host still waits
    # for the discovery lists, weak_head/loss_fused.py: early_backward; the clean rows re-attached afterwards are a
    # few hundred).  Beyond the reserve the buffers are re-allocated and the filled blocks copied.
    reserve = 0
    split = False     # this layer's backward runs on split planes although the process-wide mode's does not (Shadow.bwd2)
    # dyn.Dyn: the LENGTH of the reduction lives on the device (round 6, weak_head/loss_device.py): blocks registered by
    # CAPACITY whose live widths only the GPU knows are packed behind one another at device-side offsets; the GEMM then
    # runs over [0, *dyn_k.t) of the sum(rows) columns that exist
    dyn_k = None
    # True while another stream is still filling column blocks of this batch (loss_fused: the contrastive branch runs beside
    # the dense losses' early backward): the evaluation that completes the batch does not run the GEMM, the owner flushes
    # once both streams have met (flush_held)
    hold = False

    def __init__(self):
        self.rows, self.filled, self.dzt, self.xt, self.kpad, self.done = [], 0, None, None, 0, []

    def register(self, m):
        """Reserve the column block of an evaluation over m rows (split precision: one block per plane product)."""
        self.rows.append(_r64(m) * (len(P.patterns("gemm")[0]) if (P.bwd_split() or self.split) else 1))
        if self.dzt is not None:
            self.done.append(False)
            need = sum(self.rows)
            if need > self.dzt.shape[1]:
                step_trace.count("wgrad_batch_regrow")
                old = need - self.rows[-1]
                dzt = torch.empty((self.dzt.shape[0], need + self.reserve), dtype=torch.bfloat16, device=self.dzt.device)
                xt = torch.empty((self.xt.shape[0], need + self.reserve), dtype=torch.bfloat16, device=self.xt.device)
                dzt[:, :old].copy_(self.dzt[:, :old])
                xt[:, :old].copy_(self.xt[:, :old])
                self.dzt, self.xt = dzt, xt
            self.kpad = need
        return len(self.rows) - 1

    def offset(self, slot):
        return sum(self.rows[:slot])

    def buffers(self, n_out, k_in, device):
        if self.dzt is None:
            self.kpad = sum(self.rows)
            cap = self.kpad + self.reserve
            self.dzt = torch.empty((n_out, cap), dtype=torch.bfloat16, device=device)
            self.xt = torch.empty((k_in, cap), dtype=torch.bfloat16, device=device)
            self.done = [False] * len(self.rows)
        return self.dzt, self.xt

    def reset(self):
        self.rows, self.filled, self.dzt, self.xt, self.kpad, self.done = [], 0, None, None, 0, []
        self.dyn_k = None
        self.hold = False

    def flush(self, weight, tag=None):
        """Run the GEMM over whatever was filled (blocks of evaluations whose backward never ran are zeroed)."""
        if self.dzt is None:
            self.reset()
            return
        off = 0
        for r, d in zip(self.rows, self.done):
            if not d:                       # 0 x garbage would still be NaN: clear both operands' blocks
                self.dzt[:, off:off + r].zero_()
                self.xt[:, off:off + r].zero_()
            off += r
        fresh = weight.grad is None or getattr(weight, "_odw_fresh", False)
        if weight.grad is None:
            weight.grad = torch.empty_like(weight)
        weight._odw_fresh = False
        kernel_timer.layer = tag and tag + "_wgrad"
        n_out, k_in = weight.shape
        # data-parallel runs (engine.GradExchange): the product is cut into row blocks of the gradient -- contiguous
        # pieces of the flat gradient buffer -- and each block is handed to the exchange as it retires, so that the
        # all-reduce of fc6's 411 MB runs under the rest of this GEMM and of the backward instead of after it
        ready = getattr(weight, "_odw_grad_ready", None)
        rows = int(getattr(weight, "_odw_slice_rows", 0)) if ready is not None else 0
        tp = len(P.patterns("gemm")[0]) if (P.bwd_split() or self.split) else 1
        if self.dyn_k is not None:          # the reduction's live length is a device value (blocks packed at device offsets)
            from . import dyn as _dyn

            def product(a, m_rows, out):
                _dyn.gemm_nt(a, self.xt, m_rows, k_in, self.kpad, out, accumulate=not fresh, k=self.dyn_k, planes=tp,
                             tag=tag and tag + "_wgrad")
        else:
            def product(a, m_rows, out):
                gemm_nt(a, self.xt, m_rows, k_in, self.kpad, out, accumulate=not fresh, planes=tp)
        if rows <= 0 or rows >= n_out:
            product(self.dzt, n_out, weight.grad)
            if ready is not None:
                ready(weight, 0, n_out)
        else:
            for r0 in range(0, n_out, rows):
                r1 = min(n_out, r0 + rows)
                product(self.dzt[r0:r1], r1 - r0, weight.grad[r0:r1])
                ready(weight, r0, r1)
        kernel_timer.layer = None
        self.reset()


class _FusedLinear(torch.autograd.Function):
    """y = dropout(relu(x W^T + b)) on the matrix cores, bf16 in / bf16 or fp32 out.

    backward: one prologue kernel rebuilds the ReLU+dropout mask from the saved output and emits
    dZ / dZ^T / db; dX and dW are two more launches of the same NT GEMM; dW is accumulated
    straight into weight.grad (fp32) -- nothing the size of fc6's 411 MB gradient is ever
    materialised twice."""

    @staticmethod
    def forward(ctx, x, weight, bias, shadow, relu, drop_p, segs, out_f32, timer_tag, grad_rows=None, row_ids=None,
                grad_mode=True):
        """grad_rows = (a, b): only rows [a, b) of this evaluation take part in backward (the clean half of the
        stacked pass carries gradient for a few hundred rows only, which are re-evaluated separately);
        row_ids: x holds gathered rows of a larger pass -- their dropout draws are those of the original rows."""
        sh = shadow.refresh()
        M, K = x.shape
        N = weight.shape[0]
        xb = x if x.dtype == torch.bfloat16 else to_bf16(x)
        xb = xb if xb.stride(1) == 1 and xb.stride(0) % 8 == 0 else xb.contiguous()
        y = torch.empty((M, N), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
        kernel_timer.layer = timer_tag and timer_tag + "_fwd"
        gemm_nt(xb, sh.w, M, N, K, y, bias=bias, relu=relu, drop_p=drop_p, segs=segs, row_ids=row_ids)
        kernel_timer.layer = None
        ctx.save_for_backward(xb, y if (relu or drop_p > 0) else None, weight, bias)
        slot = None
        batch = getattr(sh, "batch", None)
        if batch is not None and weight.is_leaf and weight.requires_grad and grad_mode:    # not under torch.no_grad()
            slot = batch.register((grad_rows[1] - grad_rows[0]) if grad_rows is not None else M)
        ctx.cfg = (sh, relu, drop_p, x.dtype, timer_tag, grad_rows, slot)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, y, weight, bias = ctx.saved_tensors
        dx, dw = _backward_single_plane(xb, y, weight, bias, ctx.cfg, dy, ctx.needs_input_grad[0])
        return dx, dw, None, None, None, None, None, None, None, None, None, None


def _backward_single_plane(xb, y, weight, bias, cfg, dy, need_dx, absmax_holder=None):
    """Backward of a fused Linear with every product on ONE bf16 plane per operand ("bf16" and "bf16x2f" modes).
    xb / y are the saved input / output (bf16, or fp32 after a split-precision forward: the prologue and transpose
    kernels round to bf16 as they read); dy may be fp32 or bf16; dx comes back in the input's dtype.
    absmax_holder: the input is ROI pooling's stacked operand and its backward scatters dx in fixed point -- the input-gradient
    GEMM leaves max |dx| with the holder (`.absmax` = (word, dx's address, first row, rows)) instead of a 200 MB pre-pass."""
    sh, relu, drop_p, x_dtype, tag, grad_rows, slot = cfg
    M_all, K = xb.shape
    N = weight.shape[0]
    dy = dy.contiguous()
    if grad_rows is not None:           # rows outside [a, b) neither send nor receive gradient
        ra, rb = grad_rows[0], grad_rows[1]
        dy, xb = dy[ra:rb], xb[ra:rb]
        y = y[ra:rb] if y is not None else None
    M = xb.shape[0]
    n8, m8 = _r64(N), _r64(M)
    x_f32 = 1 if xb.dtype == torch.float32 else 0
    dz = torch.empty((M, n8), dtype=torch.bfloat16, device=dy.device)
    batch = sh.batch if slot is not None else None
    if batch is not None:                  # this evaluation's column block of the shared dZ^T / X^T matrices
        dzt_all, xt_all = batch.buffers(N, K, dy.device)
        off = batch.offset(slot)
        dzt, ld_t, t_cols = dzt_all[:, off:], dzt_all.stride(0), m8
    else:
        dzt = torch.empty((N, m8), dtype=torch.bfloat16, device=dy.device)
        ld_t, t_cols = m8, m8
    if bias is not None and bias.requires_grad:
        if not bias.is_leaf:
            raise RuntimeError("fused_linear: bias must be a leaf parameter, a constant or None")
        if bias.grad is None:
            bias.grad = torch.zeros_like(bias)
        db = bias.grad
    else:
        db = None           # no bias, or a constant one (the folded shift of a frozen batch-norm)
    scale = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
    flags = (1 if dy.dtype == torch.float32 else 0) | (2 if (y is not None and y.dtype == torch.float32) else 0)
    L.check(L.lib().odw_linear_bwd_prep_part(L.ptr(dy), flags, dy.stride(0),
                                             L.ptr(y), y.stride(0) if y is not None else 0, M, N, scale,
                                             L.ptr(dz), n8, L.ptr(dzt), ld_t, t_cols, L.ptr(db), L.stream()),
            "linear_bwd_prep")
    def input_gradient():
        if not need_dx:
            return None
        dx_all = torch.empty((M_all, K), dtype=x_dtype, device=dy.device)
        dx = dx_all
        if grad_rows is not None:
            if len(grad_rows) < 3 or grad_rows[2]:      # (a, b, False): the consumer ignores rows outside [a, b)
                dx_all[:ra].zero_()
                dx_all[rb:].zero_()
            dx = dx_all[ra:rb]

        def launch(dz=dz, dx=dx):
            kernel_timer.layer = tag and tag + "_dgrad"
            if absmax_holder is not None and dx.dtype == torch.float32 and K % 4 == 0 and dx.stride(0) == K:
                word = torch.zeros(16, dtype=torch.int32, device=dx.device)
                gemm_nt(dz, sh.wt, M, K, N, dx, absmax=word)
                zero_outside = grad_rows is None or len(grad_rows) < 3 or bool(grad_rows[2])
                absmax_holder.absmax = (word, dx_all.data_ptr(), ra if grad_rows is not None else 0, M, zero_outside)
            else:
                gemm_nt(dz, sh.wt, M, K, N, dx)
            kernel_timer.layer = None
        # A caller that runs this backward EARLY (weak_head/loss_fused.py) may ask for the large input-gradient GEMMs to be
        # handed back instead of launched: their result is not read before the very end of the step's backward (the ROI
        # pooling node), and the caller launches them right before it starts the LATE backward -- ~0.3 ms of matrix-core
        # work under which the host issues the ~100 small launches of the contrastive loss's backward, instead of the GPU
        # waiting for them one by one (profiles/r03/hip_v8_gaps.csv: 0.35 ms idle in that stretch).
        if deferred_dgrad is not None and weight is deferred_weight:
            deferred_dgrad.append(launch)
        else:
            launch()
        return dx_all

    # a weight whose gradient is exchanged as it retires (N > 1 ranks): its gradient first, so that the collective
    # also runs under this layer's input-gradient GEMM
    wgrad_first = batch is not None and getattr(weight, "_odw_grad_ready", None) is not None
    dx = None if wgrad_first else input_gradient()
    dw = None
    if weight.requires_grad and batch is not None:
        L.check(L.lib().odw_transpose_to_bf16_part(L.ptr(xb), x_f32, xb.stride(0), M, K, L.ptr(xt_all[:, off:]),
                                                   xt_all.stride(0), m8, L.stream()), "transpose_to_bf16")
        batch.done[slot] = True
        batch.filled += 1
        if batch.filled == len(batch.rows) and not batch.hold:
            batch.flush(weight, tag)
        if wgrad_first:
            dx = input_gradient()
    elif weight.requires_grad:
        xt = transpose_bf16(xb, M, K)
        if weight.is_leaf:          # accumulate straight into the parameter's gradient buffer
            fresh = weight.grad is None or getattr(weight, "_odw_fresh", False)
            if weight.grad is None:
                weight.grad = torch.empty_like(weight)
            weight._odw_fresh = False
            target = weight.grad
        else:                       # a derived weight (e.g. the concatenated predictor heads)
            fresh = True
            target = dw = torch.empty_like(weight)
        kernel_timer.layer = tag and tag + "_wgrad"
        gemm_nt(dzt, xt, N, K, M, target, accumulate=not fresh)
        kernel_timer.layer = None
    return dx, dw


class _SplitLinear(torch.autograd.Function):
    """The same fused Linear in a split precision mode (precision.py: "bf16x3" = fp32-grade): x, y and every gradient
    are fp32 tensors; right before each of the three products (forward, input gradient, weight gradient) the two
    operands are laid out as bf16 planes along the reduction axis (csrc/split.hip) and the unchanged MFMA GEMM runs
    over K' = T * K with the same fused epilogue (bias, ReLU, counter-based dropout, accumulate)."""

    @staticmethod
    def forward(ctx, x, weight, bias, shadow, relu, drop_p, segs, timer_tag, grad_rows, row_ids, grad_mode, planes=None,
                planes_cm=None):
        """planes: the operand already laid out as bf16 planes (M x T*r64(K), pattern A of the mode) by its producer
        (ROI pooling in "bf16x2f"); x is then only the autograd handle of that operand (see planes_handle).
        A weight kept as cell-major planes (Shadow.w_cm: the first head Linear in "bf16x2f") takes its operand the same
        way -- planes_cm (M x 2K) from the producer, else x is split here -- and `planes` then only carries the hi plane
        the backward reads."""
        sh = shadow.refresh()
        pa, pb = P.patterns("gemm")
        T = len(pa)
        M, K = x.shape
        N = weight.shape[0]
        kp = _r64(K)
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        if sh.w_cm is not None:
            C, S = sh.cm
            if planes_cm is not None:
                assert planes is not None and planes.shape[0] == M and planes.shape[1] >= K and planes_cm.shape == (M, 2 * K)
                xs, x32 = planes_cm, planes[:, :K]
            else:
                if planes is not None:
                    raise RuntimeError("fused_linear: this weight is kept as cell-major planes; an operand laid out as "
                                       "channel-major planes cannot meet it")
                x32 = x if x.dtype == torch.float32 else x.float()
                x32 = x32 if x32.stride(1) == 1 else x32.contiguous()
                xs = split_rows_cm(x32, C, S)
            kernel_timer.layer = timer_tag and timer_tag + "_fwd"
            gemm_nt_cm(xs, sh.w_cm, M, N, C, S, y, bias=bias, relu=relu, drop_p=drop_p, segs=segs, row_ids=row_ids)
            kernel_timer.layer = None
        else:
            if planes is not None:
                assert planes.shape == (M, T * kp) and planes.dtype == torch.bfloat16 and planes.stride(1) == 1
                xs, x32 = planes, planes[:, :K]          # backward reads the hi plane (bf16, row stride T*kp)
            else:
                x32 = x if x.dtype == torch.float32 else x.float()
                x32 = x32 if x32.stride(1) == 1 else x32.contiguous()
                xs = P.split_rows(x32, pa, kp)
            kernel_timer.layer = timer_tag and timer_tag + "_fwd"
            gemm_nt(xs, sh.w, M, N, T * kp, y, bias=bias, relu=relu, drop_p=drop_p, segs=segs, row_ids=row_ids, planes=T)
            kernel_timer.layer = None
        del xs
        ctx.save_for_backward(x32, y if (relu or drop_p > 0) else None, weight, bias)
        slot = None
        batch = getattr(sh, "batch", None)
        if batch is not None and weight.is_leaf and weight.requires_grad and grad_mode:
            slot = batch.register((grad_rows[1] - grad_rows[0]) if grad_rows is not None else M)
        ctx.cfg = (sh, relu, drop_p, x.dtype, timer_tag, grad_rows, slot)
        return y

    @staticmethod
    def backward(ctx, dy):
        x32, y, weight, bias = ctx.saved_tensors
        dx, dw = _backward_split(x32, y, weight, bias, ctx.cfg, dy, ctx.needs_input_grad[0])
        return dx, dw, None, None, None, None, None, None, None, None, None, None, None


def _backward_split(x32, y, weight, bias, cfg, dy, need_dx):
    """Backward of a fused Linear with every product on split planes (modes "bf16x3" / "bf16x2"; Shadow.bwd2 layers of
    "bf16x2f"): x32 / y are the saved fp32 input / output."""
    if True:
        if x32.dtype != torch.float32:
            raise RuntimeError("_SplitLinear: a pre-split operand is only supported by the single-plane backward (bf16x2f)")
        sh, relu, drop_p, x_dtype, tag, grad_rows, slot = cfg
        pa, pb = P.patterns("gemm")
        T = len(pa)
        M_all, K = x32.shape
        N = weight.shape[0]
        dy = dy if dy.dtype == torch.float32 else dy.float()
        dy = dy if dy.stride(1) == 1 else dy.contiguous()
        if grad_rows is not None:
            ra, rb = grad_rows[0], grad_rows[1]
            dy, x32 = dy[ra:rb], x32[ra:rb]
            y = y[ra:rb] if y is not None else None
        M = x32.shape[0]
        np_, m64 = _r64(N), _r64(M)
        if bias is not None and bias.requires_grad:
            if not bias.is_leaf:
                raise RuntimeError("fused_linear: bias must be a leaf parameter, a constant or None")
            if bias.grad is None:
                bias.grad = torch.zeros_like(bias)
            db = bias.grad
        else:
            db = None
        scale = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
        dz = P.bwd_mask(dy, y, scale, db)
        dx = None
        if need_dx:
            dx_all = torch.empty((M_all, K), dtype=torch.float32, device=dy.device)
            dx = dx_all
            if grad_rows is not None:
                if len(grad_rows) < 3 or grad_rows[2]:
                    dx_all[:ra].zero_()
                    dx_all[rb:].zero_()
                dx = dx_all[ra:rb]
            dzs = P.split_rows(dz, pa, np_)
            kernel_timer.layer = tag and tag + "_dgrad"
            gemm_nt(dzs, sh.wt, M, K, T * np_, dx, planes=T)
            kernel_timer.layer = None
            del dzs
            dx = dx_all if x_dtype == torch.float32 else dx_all.to(x_dtype)
        dw = None
        batch = sh.batch if slot is not None else None
        if weight.requires_grad and batch is not None:
            dzt_all, xt_all = batch.buffers(N, K, dy.device)
            off = batch.offset(slot)
            P.split_cols(dz, pa, m64, out=dzt_all[:, off:])
            P.split_cols(x32, pb, m64, out=xt_all[:, off:])
            batch.done[slot] = True
            batch.filled += 1
            if batch.filled == len(batch.rows):
                batch.flush(weight, tag)
        elif weight.requires_grad:
            dzt = P.split_cols(dz, pa, m64)
            xt = P.split_cols(x32, pb, m64)
            if weight.is_leaf:
                fresh = weight.grad is None or getattr(weight, "_odw_fresh", False)
                if weight.grad is None:
                    weight.grad = torch.empty_like(weight)
                weight._odw_fresh = False
                target = weight.grad
            else:
                fresh = True
                target = dw = torch.empty_like(weight)
            kernel_timer.layer = tag and tag + "_wgrad"
            gemm_nt(dzt, xt, N, K, T * m64, target, accumulate=not fresh, planes=T)
            kernel_timer.layer = None
        return dx, dw


class _MixedLinear(torch.autograd.Function):
    """Precision mode "bf16x2f": the forward product of _SplitLinear (two bf16 planes per operand, three plane
    products, fp32 in / fp32 out -- what the loss values and every index selection depend on), the backward of
    _FusedLinear (dZ, X and W each rounded to one bf16 plane; dX comes back as fp32)."""

    forward = staticmethod(_SplitLinear.forward)

    @staticmethod
    def backward(ctx, dy):
        x32, y, weight, bias = ctx.saved_tensors
        sh, relu, drop_p, x_dtype, tag, grad_rows, slot = ctx.cfg
        if sh.bwd2 and x32.dtype == torch.float32:      # (measurement mode ODW_BWD2: this layer's backward on two planes)
            dx, dw = _backward_split(x32, y, weight, bias, ctx.cfg, dy, ctx.needs_input_grad[0])
            return dx, dw, None, None, None, None, None, None, None, None, None, None, None
        cfg = (sh, relu, drop_p, torch.float32, tag, grad_rows, slot)
        dx, dw = _backward_single_plane(x32, y, weight, bias, cfg, dy, ctx.needs_input_grad[0])
        if dx is not None and x_dtype != torch.float32:
            dx = dx.to(x_dtype)
        return dx, dw, None, None, None, None, None, None, None, None, None, None, None


class _PairLinear(torch.autograd.Function):
    """fc6 of the clean AND the DropBlock pass of ROIWeakRegHead.forward (weak_head.py:107-112) from ONE sweep over the
    clean operand ("bf16x2f"; csrc/gemm_bf16.hip: gemm_nt_cm_kernel): y (2P x N) = [clean rows; DropBlock rows], the values
    of the stacked evaluation it replaces (other summation order).  x is the autograd handle of the stacked (2P x K)
    operand; planes_cm (P x 2K) the clean rows as cell-major planes, planes_bwd (2P x >= K) what the single-plane backward
    reads (= _MixedLinear's): the bf16 hi plane of both halves in the reference's order (ROI pooling writes it), or the
    stacked fp32 operand itself."""

    @staticmethod
    def forward(ctx, x, weight, bias, shadow, planes_cm, planes_bwd, keep, keep_sum, relu, drop_p, segs, timer_tag, grad_rows,
                grad_mode):
        sh = shadow.refresh()
        if sh.cm is None or sh.w_cm is None:
            raise RuntimeError("pair_linear: this weight keeps no cell-major planes (gemm.Shadow.cm)")
        C, S = sh.cm
        M2, K = x.shape
        Pn = M2 // 2
        N = weight.shape[0]
        assert planes_cm.shape == (Pn, 2 * K) and planes_bwd.shape[0] == M2 and planes_bwd.shape[1] >= K and K == C * S
        assert keep.numel() == Pn * S and keep.dtype == torch.float32 and keep.is_contiguous()
        assert segs is None or (len(segs) == 2 and segs[0][0] == 0 and segs[1][0] == Pn)
        y = torch.empty((M2, N), dtype=torch.float32, device=x.device)
        kernel_timer.layer = timer_tag and timer_tag + "_fwd"
        gemm_nt_cm(planes_cm, sh.w_cm, Pn, N, C, S, y, bias=bias, relu=relu, drop_p=drop_p, segs=segs, keep=keep,
                   keep_sum=keep_sum, drop_row0=Pn)
        kernel_timer.layer = None
        ctx.save_for_backward(planes_bwd[:, :K], y if (relu or drop_p > 0) else None, weight, bias)
        slot = None
        batch = getattr(sh, "batch", None)
        if batch is not None and weight.is_leaf and weight.requires_grad and grad_mode:
            slot = batch.register((grad_rows[1] - grad_rows[0]) if grad_rows is not None else M2)
        ctx.cfg = (sh, relu, drop_p, x.dtype, timer_tag, grad_rows, slot)
        ctx.absmax_holder = getattr(x, "_odw_absmax_holder", None)      # (set by ROI pooling's stacked forward on its handle)
        return y

    @staticmethod
    def backward(ctx, dy):
        x16, y, weight, bias = ctx.saved_tensors
        sh, relu, drop_p, x_dtype, tag, grad_rows, slot = ctx.cfg
        cfg = (sh, relu, drop_p, torch.float32, tag, grad_rows, slot)
        dx, dw = _backward_single_plane(x16, y, weight, bias, cfg, dy, ctx.needs_input_grad[0],
                                        absmax_holder=ctx.absmax_holder if x_dtype == torch.float32 else None)
        if dx is not None and x_dtype != torch.float32:
            dx = dx.to(x_dtype)
        return (dx, dw) + (None,) * 12


def pair_linear(x, weight, bias, shadow, planes_cm, planes_bwd, keep, keep_sum, relu=False, drop_p=0.0, segs=None, tag=None,
                grad_rows=None):
    """The clean + DropBlock evaluation of the first head Linear from one sweep (see _PairLinear); "bf16x2f" only."""
    if P.get_precision() != "bf16x2f":
        raise RuntimeError("pair_linear: precision mode %r (the shared forward exists for \"bf16x2f\")" % P.get_precision())
    return _PairLinear.apply(x, weight, bias, shadow, planes_cm, planes_bwd, keep, keep_sum, relu, drop_p, segs, tag, grad_rows,
                             torch.is_grad_enabled())


class _ReuseLinear(torch.autograd.Function):
    """A Linear evaluation whose OUTPUT already exists: rows `rows` of `y_full`, computed by an earlier, larger evaluation
    of the same layer on the same inputs with the same dropout draws and no autograd (the clean half of the stacked
    fc6 / fc7 / Sim_Net pass).  forward = a row gather (no GEMM); backward = the layer's ordinary single-plane backward
    over those rows (input gradient, weight-gradient batch slot, bias gradient, ReLU / dropout mask from the gathered
    output).  Round 2 RE-EVALUATED these rows (gather -> fc6 -> fc7 -> Sim_Net: three weight-streaming GEMMs of a few
    hundred rows, 0.5 ms of the bf16x2f step); the values are the same bits either way."""

    @staticmethod
    def forward(ctx, x, weight, bias, shadow, y_full, rows, relu, drop_p, timer_tag, planes):
        sh = shadow.refresh()
        M, K = x.shape
        N = weight.shape[0]
        assert y_full.shape[1] == N and rows.numel() == M
        y = y_full.index_select(0, rows.long() if rows.dtype != torch.int64 and rows.dtype != torch.int32 else rows)
        xb = planes[:, :K] if planes is not None else (x if x.stride(1) == 1 else x.contiguous())
        ctx.save_for_backward(xb, y if (relu or drop_p > 0) else None, weight, bias)
        slot = None
        batch = getattr(sh, "batch", None)
        if batch is not None and weight.is_leaf and weight.requires_grad:
            slot = batch.register(M)
        ctx.cfg = (sh, relu, drop_p, x.dtype, timer_tag, None, slot)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, y, weight, bias = ctx.saved_tensors
        if ctx.cfg[0].bwd2 and xb.dtype == torch.float32:      # (measurement mode ODW_BWD2: this layer's backward on two planes)
            dx, dw = _backward_split(xb, y, weight, bias, ctx.cfg, dy, ctx.needs_input_grad[0])
        else:
            dx, dw = _backward_single_plane(xb, y, weight, bias, ctx.cfg, dy, ctx.needs_input_grad[0])
        return dx, dw, None, None, None, None, None, None, None, None


def reuse_linear(x, weight, bias, shadow, y_full, rows, relu=False, drop_p=0.0, tag=None):
    """Rows `rows` of an evaluation of this Linear that already ran without autograd, re-attached to the graph (see
    _ReuseLinear).  Single-plane backward only ("bf16", "bf16x2f")."""
    if P.bwd_split():
        raise RuntimeError("reuse_linear: the split-precision backward keeps no single-plane operands")
    if shadow.bwd2 and x.dtype != torch.float32:
        raise RuntimeError("reuse_linear: a two-plane backward (ODW_BWD2) needs the fp32 input of the re-attached rows")
    return _ReuseLinear.apply(x, weight, bias, shadow, y_full, rows, relu, drop_p, tag, getattr(x, "_odw_planes", None))


def planes_handle(device, rows, cols):
    """The autograd stand-in of an operand that exists only as bf16 planes: a (rows x cols) fp32 tensor of zero strides
    (4 bytes of storage) that carries the graph edge and the gradient's shape / dtype.  Its producer (an autograd
    Function) returns it and attaches the planes as `_odw_planes`; fused_linear() picks them up; nothing ever reads
    the handle's values (never call .contiguous() on one: that would materialise rows x cols zeros)."""
    z = _HANDLE_ZERO.get(str(device))
    if z is None:           # one 4-byte zero per device for every handle (was a fill launch per handle, ~5 us each)
        z = _HANDLE_ZERO[str(device)] = torch.zeros(1, dtype=torch.float32, device=device)
    return z.expand(rows, cols)


_HANDLE_ZERO = {}


def fused_linear(x, weight, bias, shadow, relu=False, drop_p=0.0, segs=None, out_f32=False, tag=None, grad_rows=None,
                 row_ids=None):
    """dropout(relu(x W^T + b)) on the matrix cores.  bf16 mode: bf16 operands, bf16 (or, out_f32, fp32) result;
    split precision modes: fp32 in, fp32 out (precision.py)."""
    if P.split_mode():
        fn = _SplitLinear if P.bwd_split() else _MixedLinear
        return fn.apply(x, weight, bias, shadow, relu, drop_p, segs, tag, grad_rows, row_ids, torch.is_grad_enabled(),
                        getattr(x, "_odw_planes", None), getattr(x, "_odw_planes_cm", None))
    return _FusedLinear.apply(x, weight, bias, shadow, relu, drop_p, segs, out_f32, tag, grad_rows, row_ids,
                              torch.is_grad_enabled())      # (grad mode is always off INSIDE Function.forward)
