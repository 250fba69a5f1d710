"""Shared body of the two ROI feature extractors on the hot path -- Pooler, two Linear+ReLU+Dropout layers, and
the DropBlock / noise views the contrastive loss asks for (wetectron/modeling/backbone/vgg16.py:107-180 and
roi_heads/box_head/roi_box_feature_extractors.py:13-122 are the same code apart from the layer widths and the
positions of the Linear layers inside `classifier`).

`rand` (a DeviceRand) carries the counter-based streams; every method draws in the reference's order."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..dropblock import DropBlock2D
from ..poolers import Pooler
from ... import _lib as L
from ... import gemm
from ... import precision
from ...layers.linear import Linear
from ...utils.kernel_timer import kernel_timer


class _StackCleanAug(torch.autograd.Function):
    """pooled (P,C,h,w) fp32 + DropBlock keep mask -> the (2P x C*h*w) operand of the first GEMM (rows 0..P-1 the
    clean features, rows P..2P-1 the DropBlock view), one pass each way (csrc/head_aux.hip); bf16, or fp32 in a
    split precision mode (the GEMM front end lays it out as bf16 planes)."""

    @staticmethod
    def forward(ctx, pooled, block, block_sum, holder):
        P, C, h, w = pooled.shape
        S = h * w
        pooled = pooled.contiguous()
        out = torch.empty((2 * P, C * S), dtype=precision.act_dtype(), device=pooled.device)
        fn = L.lib().odw_stack_clean_aug_f32 if out.dtype == torch.float32 else L.lib().odw_stack_clean_aug
        L.check(fn(L.ptr(pooled), L.ptr(block), L.ptr(block_sum), P, C, S, L.ptr(out), out.stride(0), L.stream()),
                "stack_clean_aug")
        ctx.save_for_backward(block, block_sum)
        ctx.shape = (P, C, h, w)
        ctx.holder = holder
        return out

    @staticmethod
    def backward(ctx, dx):
        block, block_sum = ctx.saved_tensors
        P, C, h, w = ctx.shape
        dx = dx if dx.stride(1) == 1 else dx.contiguous()
        dp = torch.empty((P, C, h, w), dtype=torch.float32, device=dx.device)
        L.check(L.lib().odw_unstack_clean_aug_bwd(L.ptr(dx), 1 if dx.dtype == torch.float32 else 0, dx.stride(0),
                                                  L.ptr(block), L.ptr(block_sum), P, C, h * w, L.ptr(dp), L.stream()),
                "unstack_clean_aug_bwd")
        if ctx.holder is not None:          # gradients of the sampled-row views parked by _RowViews.backward
            for fold in ctx.holder.pending:
                fold(dp, False)
            ctx.holder.pending = []
            ctx.holder.done = True
        return dp, None, None, None


class _GradHolder(object):
    """Where the gradient of `pooled` is assembled for one step: the sampled-row views (a later, smaller autograd
    node that the engine runs first) park their contribution here and the stacked node folds it into the one dense
    tensor it produces anyway -- instead of a second dense (P,C,7,7) tensor, a fill and a 600 MB add."""

    def __init__(self, kind="dense"):
        # kind "dense": folds accumulate into the (P,C,h,w) fp32 gradient of `pooled` (at row_base + rows);
        # kind "extra": there is no pooled tensor (ROI pooling writes the stacked operand itself) -- folds fill an
        #               (E, C*h*w) fp32 side buffer, one row per sampled entry, roi_index[e] = the ROI it belongs to
        self.kind, self.pending, self.done, self.roi_index = kind, [], False, None
        # (extra buffer (E_cap, C*h*w) fp32, entry -> ROI list, E_cap, dyn.Dyn of the live entry count): the side buffer as the
        # device-resident contrastive branch leaves it (weak_head/loss_device.py) -- filled, its length on the device
        self.dyn_extra = None
        self.grad_out = None        # where the pooling node writes d(feature map) (a HIP-graphed body: its static input buffer)
        self.clean_rows = None      # _PoolStack: number of leading rows of dX whose gradient is meaningful (None = all)
        self.absmax = None          # (word, address of dX, first row, rows, rest zeroed): max |dX| left by the GEMM that wrote dX


_IOTA = {}


def _iota(n, device):
    """int32 0..n-1 on the device (cached, grown geometrically): identity row lists for the side buffer."""
    t = _IOTA.get(str(device))
    if t is None or t.numel() < n:
        t = _IOTA[str(device)] = torch.arange(max(n, 4096), dtype=torch.int32, device=device)
    return t[:n]


class _RowViews(torch.autograd.Function):
    """src[rows] of every (image, class) -> drop view and noise view, stacked as the bf16 GEMM operand
    (csrc/head_aux.hip: rows_drop_noise_kernel); groups = [(row_base, rows int32 device view, k, keys...)];
    src = the fp32 pooled tensor (P,C,h,w), or the bf16 stacked operand of _PoolStack (its first P rows)."""

    @staticmethod
    def forward(ctx, src, groups, holder, gamma, S, values=None, src_cm=None):
        """values: the fp32 pooled tensor to read when `src` is only the autograd handle of a planes operand
        (_PoolStackPlanes: its gradient is parked in `holder` like that of the bf16 stacked operand).
        src_cm ("bf16x2f", shared clean + DropBlock forward): the clean rows as cell-major planes (R x 2K, [hi | mid]) -- the
        views are read from them and come back as (handle, channel-major hi plane, cell-major planes) of the 2 x total view
        rows (csrc/head_aux.hip: rows_views_cm_kernel): the operand of the first head Linear, no fp32 views, no split pass."""
        if src_cm is not None:
            CS = src.shape[1]
            C = CS // S
            assert src_cm.dim() == 2 and src_cm.shape[1] == 2 * CS and src_cm.dtype == torch.bfloat16 and src_cm.stride(1) == 1
            total = sum(g[2] for g in groups)
            out_cm = torch.empty((2 * total, 2 * CS), dtype=torch.bfloat16, device=src_cm.device)
            out_hi = torch.empty((2 * total, CS), dtype=torch.bfloat16, device=src_cm.device)
            sums = torch.empty(len(groups), dtype=torch.float32, device=src_cm.device)
            lib, st, row0 = L.lib(), L.stream(), 0
            for gi, (base, rows, k, kd, kn) in enumerate(groups):
                L.check(lib.odw_rows_views_cm(L.ptr(src_cm), src_cm.stride(0), CS, L.ptr(rows), base, k, C, S, gamma, kd[0], kd[1],
                                              kn[0], kn[1], L.ptr(sums[gi:]), L.ptr(out_cm), out_cm.stride(0), CS, L.ptr(out_hi),
                                              out_hi.stride(0), row0, st), "rows_views_cm")
                row0 += 2 * k
            ctx.args = (groups, holder, gamma, tuple(src.shape), sums, C, S, False)
            ctx.set_materialize_grads(False)
            ctx.mark_non_differentiable(out_hi, out_cm)
            return gemm.planes_handle(src_cm.device, 2 * total, CS), out_hi, out_cm
        if values is not None:
            src = values
        src = src.contiguous()
        from_bf16 = src.dtype == torch.bfloat16
        CS = src.shape[1] if from_bf16 else src.shape[1] * src.shape[2] * src.shape[3]
        C = CS // S
        total = sum(g[2] for g in groups)
        out = torch.empty((2 * total, CS), dtype=precision.act_dtype(), device=src.device)
        sums = torch.empty(len(groups), dtype=torch.float32, device=src.device)
        lib, st, row0 = L.lib(), L.stream(), 0
        if out.dtype == torch.float32 and from_bf16:
            raise RuntimeError("_RowViews: a split precision mode reads the fp32 pooled tensor, not a bf16 operand")
        for gi, (base, rows, k, kd, kn) in enumerate(groups):
            if out.dtype == torch.float32:
                L.check(lib.odw_rows_drop_noise_f32(L.ptr(src), L.ptr(rows), base, k, C, S, gamma, kd[0], kd[1], kn[0],
                                                    kn[1], L.ptr(sums[gi:]), L.ptr(out), out.stride(0), row0, st),
                        "rows_drop_noise_f32")
            else:
                L.check(lib.odw_rows_drop_noise(L.ptr(src), 1 if from_bf16 else 0, L.ptr(rows), base, k, C, S, gamma,
                                                kd[0], kd[1], kn[0], kn[1], L.ptr(sums[gi:]), L.ptr(out), out.stride(0),
                                                row0, st), "rows_drop_noise")
            row0 += 2 * k
        ctx.args = (groups, holder, gamma, tuple(src.shape), sums, C, S, from_bf16)
        return out

    @staticmethod
    def backward(ctx, dx, *unused):
        groups, holder, gamma, shape, sums, C, S, from_bf16 = ctx.args
        dx = dx if dx.stride(1) == 1 else dx.contiguous()
        f32 = 1 if dx.dtype == torch.float32 else 0

        def fold(target, identity_rows, fresh=False):
            """accumulate d(src rows) into `target`: the dense fp32 gradient of pooled at the sampled rows, or
            (identity_rows) the (E, C*S) side buffer at consecutive entries; fresh: the side buffer is uninitialised and
            every entry is written by exactly one fold (a store instead of a read-add-write of a zero-filled buffer)"""
            lib, st, row0, e0 = L.lib(), L.stream(), 0, 0
            fn = lib.odw_rows_drop_noise_bwd_store if (fresh and identity_rows) else lib.odw_rows_drop_noise_bwd
            for gi, (base, rows, k, kd, kn) in enumerate(groups):
                r = _iota(k, dx.device) if identity_rows else rows
                L.check(fn(L.ptr(dx), f32, dx.stride(0), row0, L.ptr(r), e0 if identity_rows else base, k, C, S, gamma, kd[0], kd[1],
                           kn[0], kn[1], L.ptr(sums[gi:]), L.ptr(target), st), "rows_drop_noise_bwd")
                row0 += 2 * k
                e0 += k

        fold.entries = sum(g[2] for g in groups)         # side-buffer entries [0, entries) are this fold's alone
        fold.interval = (0, fold.entries)

        if holder is not None and not holder.done:
            holder.pending.append(fold)          # the stacked node has not produced its gradient yet: it folds this in
            return None, None, None, None, None, None, None
        if from_bf16:      # (never taken in the training step) dense gradient of the stacked operand itself
            d = torch.zeros((shape[0], C * S), dtype=torch.float32, device=dx.device)
            fold(d, False)
            return d.to(torch.bfloat16), None, None, None, None, None, None
        dp = torch.zeros(shape, dtype=torch.float32, device=dx.device)
        fold(dp, False)
        return dp, None, None, None, None, None, None


class _RowGather(torch.autograd.Function):
    """rows of the stacked bf16 operand as a new (R, C*S) operand; the gradient of those rows is parked for the
    pooling node (entries [e0, e0+R) of its fp32 side buffer)."""

    @staticmethod
    def forward(ctx, stacked, rows, holder, e0, planes=None, planes_cm=None):
        """planes: `stacked` is the handle of a planes operand (_PoolStackPlanes) -- its rows are gathered from the
        planes and come back as a new handle (the caller attaches the gathered planes, returned second; planes_cm: the
        cell-major forward planes of the same rows, returned third)."""
        ctx.args = (holder, e0, int(rows.numel()))
        if planes is not None:
            picked = planes.index_select(0, rows)
            ctx.set_materialize_grads(False)        # (or autograd hands backward a zero tensor the size of `picked`)
            handle = gemm.planes_handle(planes.device, int(rows.numel()), stacked.shape[1])
            if planes_cm is not None:
                picked_cm = planes_cm.index_select(0, rows)
                ctx.mark_non_differentiable(picked, picked_cm)
                return handle, picked, picked_cm
            ctx.mark_non_differentiable(picked)
            return handle, picked
        return stacked.index_select(0, rows)

    @staticmethod
    def backward(ctx, dx, *unused):
        holder, e0, n = ctx.args

        def fold(extra, identity_rows, fresh=False):
            extra[e0:e0 + n].copy_(dx)       # the entries are this node's alone (zero before): a converting copy, not
                                             # the mixed-dtype add_ (104 us for 250 rows against ~10)

        fold.entries = n
        fold.interval = (e0, n)

        if holder is None or holder.done:
            raise RuntimeError("_RowGather: the pooling node already ran its backward")
        holder.pending.append(fold)
        return None, None, None, None, None, None


class _PoolStack(torch.autograd.Function):
    """ROIPool of the feature map written directly as the stacked bf16 operand of the first head GEMM (rows 0..R-1
    the pooled features, rows R..2R-1 their DropBlock view) with a 16-bit argmax; backward scatters the gradient of
    both halves plus the parked gradients of the sampled-row views into d(features) (csrc/roi_pool.hip)."""

    @staticmethod
    def forward(ctx, feat, rois5, keep, keep_sum, holder, scale, ph, pw, nhwc=None):
        feat = feat.contiguous()
        rois5 = rois5.contiguous().float()
        B, C, H, W = feat.shape
        R, nb = rois5.shape[0], ph * pw
        x = torch.empty((2 * R, C * nb), dtype=torch.bfloat16, device=feat.device)
        argmax = torch.empty((R, C * nb), dtype=torch.int16, device=feat.device)
        lib = L.lib()
        ws_bytes = lib.odw_roi_pool_workspace(R, ph, pw)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=feat.device)
        if nhwc is not None and ph == 7 and pw == 7 and C % 64 == 0 and nhwc.numel() == feat.numel():
            # the backbone's own NHWC bf16 map (`feat` is its fp32 NCHW copy): (ROI, 64-channel) workgroups
            ws_bytes = lib.odw_roi_pool_stack_nhwc_workspace(R, B, C, H, W)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=feat.device)
            # alg = SURVEY 8(d)'s bytes of the reference operator this replaces (fp32 out + int32 argmax written, map read once)
            with kernel_timer.region("roi_pool_stack_fwd_nhwc", nbytes=float(B * C * H * W * 2 + 2 * R * C * nb * 2 + R * C * nb * 2),
                                     alg=float(2 * R * C * nb * 4 + B * C * H * W * 4)):
                L.check(lib.odw_roi_pool_stack_forward_nhwc(L.ptr(nhwc), L.ptr(rois5), scale, B, C, H, W, R, L.ptr(keep),
                                                            L.ptr(keep_sum), L.ptr(x), x.stride(0), L.ptr(argmax), L.ptr(ws),
                                                            ws_bytes, L.stream()), "roi_pool_stack_forward_nhwc")
        else:
            L.check(lib.odw_roi_pool_stack_forward(L.ptr(feat), L.ptr(rois5), scale, B, C, H, W, R, ph, pw, L.ptr(keep),
                                                   L.ptr(keep_sum), L.ptr(x), x.stride(0), L.ptr(argmax), L.ptr(ws),
                                                   ws_bytes, L.stream()), "roi_pool_stack_forward")
        ctx.save_for_backward(rois5, keep, keep_sum, argmax)
        ctx.dims = (B, C, H, W, R, ph, pw)
        ctx.holder = holder
        return x

    @staticmethod
    def backward(ctx, dx):
        rois5, keep, keep_sum, argmax = ctx.saved_tensors
        B, C, H, W, R, ph, pw = ctx.dims
        dx = dx if dx.stride(1) == 1 else dx.contiguous()
        holder = ctx.holder
        extra, roi_index, E = None, None, 0
        dyn_extra = getattr(holder, "dyn_extra", None) if holder is not None else None
        if dyn_extra is not None and holder.pending:
            raise RuntimeError("_PoolStack: both a device-resident side buffer and parked host-shaped folds")
        if holder is not None and holder.pending:
            roi_index = holder.roi_index
            if isinstance(roi_index, (list, tuple)):
                roi_index = roi_index[0] if len(roi_index) == 1 else torch.cat(list(roi_index))
            E = int(roi_index.numel())
            # every entry written by exactly one parked fold (the usual case): no zero fill of the 45-90 MB buffer.  Checked on
            # the folds' INTERVALS, not their sizes: two folds over the same entries with another range missing would add up to
            # E as well, leave uninitialised rows and lose the overlapped contribution (the STORE form does not accumulate)
            spans = sorted(getattr(f, "interval", (-1, 0)) for f in holder.pending)
            pos = 0
            for a, n in spans:
                if a != pos or n <= 0:
                    pos = -1
                    break
                pos = a + n
            fresh = pos == E and os.environ.get("ODW_EXTRA_ZEROS") != "1"
            extra = (torch.empty if fresh else torch.zeros)((E, C * ph * pw), dtype=torch.float32, device=dx.device)
            for fold in holder.pending:
                fold(extra, True, fresh)
            holder.pending = []
        if holder is not None:
            holder.done = True
        dfeat = getattr(holder, "grad_out", None) if holder is not None else None     # the body graph's own input buffer
        if dfeat is None or tuple(dfeat.shape) != (B, C, H, W) or dfeat.dtype != torch.float32 or not dfeat.is_contiguous():
            dfeat = torch.empty((B, C, H, W), dtype=torch.float32, device=dx.device)
        skip_clean = 1 if (holder is not None and holder.clean_rows == 0) else 0      # sparse backward: clean half unset
        ws = torch.empty(64, dtype=torch.uint8, device=dx.device)       # the launch's fixed-point scale (odw_fixed.h)
        K = C * ph * pw
        # max |dx| left by the GEMM that produced dx (gemm._backward_single_plane): valid for THIS tensor and the rows this
        # launch reads -- anything else (autograd summed two gradients into a new tensor, other rows) takes the pre-pass
        pre = getattr(holder, "absmax", None) if holder is not None else None
        if holder is not None:
            holder.absmax = None
        scaled = False
        if pre is not None and dx.dtype == torch.float32 and dx.stride(0) == K:
            word, ptr, row0, rows, zero_outside = pre
            covers = (row0 == R and rows == R) if skip_clean else ((row0 == 0 and rows == 2 * R) or zero_outside)
            scaled = ptr == dx.data_ptr() and covers and tuple(dx.shape) == (2 * R, K)
            if scaled:
                ws = word
        nbytes = float((R if skip_clean else 2 * R) * K * dx.element_size() + R * K * 2 + E * K * 4 + B * C * H * W * 4)
        if dyn_extra is not None:
            extra, roi_index, e_cap, e_dyn = dyn_extra
            holder.dyn_extra = None
            nbytes += float(e_dyn.hint * K * 4)
            if scaled:
                nbytes -= float(R * K * 4)          # (the pre-pass's read of dx)
            fn = L.lib().odw_roi_pool_stack_backward_scaled if scaled else L.lib().odw_roi_pool_stack_backward_dyn
            with kernel_timer.region("roi_pool_stack_backward", nbytes=nbytes, alg=float(2 * R * K * 4 + B * C * H * W * 4)):
                L.check(fn(L.ptr(dx), 1 if dx.dtype == torch.float32 else 0, dx.stride(0),
                           L.ptr(argmax), L.ptr(rois5), L.ptr(keep), L.ptr(keep_sum),
                           L.ptr(extra), L.ptr(roi_index), e_cap, L.ptr(e_dyn.t), skip_clean,
                           B, C, H, W, R, ph, pw, L.ptr(dfeat), L.ptr(ws), 64, L.stream()),
                        "roi_pool_stack_backward_dyn")
            return dfeat, None, None, None, None, None, None, None, None
        with kernel_timer.region("roi_pool_stack_backward", nbytes=nbytes, alg=float(2 * R * K * 4 + B * C * H * W * 4)):
            if scaled:
                L.check(L.lib().odw_roi_pool_stack_backward_scaled(L.ptr(dx), 1, dx.stride(0), L.ptr(argmax), L.ptr(rois5),
                                                                   L.ptr(keep), L.ptr(keep_sum), L.ptr(extra), L.ptr(roi_index),
                                                                   E, None, skip_clean, B, C, H, W, R, ph, pw, L.ptr(dfeat),
                                                                   L.ptr(ws), 64, L.stream()), "roi_pool_stack_backward_scaled")
            else:
                L.check(L.lib().odw_roi_pool_stack_backward_ws(L.ptr(dx), 1 if dx.dtype == torch.float32 else 0, dx.stride(0),
                                                               L.ptr(argmax), L.ptr(rois5), L.ptr(keep), L.ptr(keep_sum),
                                                               L.ptr(extra), L.ptr(roi_index), E, skip_clean, B, C, H, W, R, ph,
                                                               pw, L.ptr(dfeat), L.ptr(ws), 64, L.stream()),
                        "roi_pool_stack_backward")
        return dfeat, None, None, None, None, None, None, None, None


class _PoolStackPlanes(torch.autograd.Function):
    """_PoolStack in the precision mode "bf16x2f": ROIPool of the fp32 NHWC feature map written directly as the bf16
    PLANES of the stacked operand of the first head GEMM (csrc/roi_pool.hip: roi_pool_stack_fwd_nhwc_f32), with the
    16-bit argmax and the fp32 pooled values (the sampled-row views read those).  Returns (handle, planes, pooled32):
    the handle is the autograd stand-in of the (2R x C*49) operand (gemm.planes_handle).  Backward = _PoolStack's."""

    @staticmethod
    def forward(ctx, feat, nhwc32, rois5, keep, keep_sum, holder, scale, ph, pw, pair=False):
        """pair: the layout of the shared clean + DropBlock forward (gemm.pair_linear) -- `planes` carries only what the
        backward reads (the hi plane of both halves, 2R x K) and a fourth result holds the clean rows as the two
        cell-major planes (R x 2K) the forward sweeps: 4 plane-rows per ROI written instead of 6."""
        rois5 = rois5.contiguous().float()
        B, C, H, W = feat.shape
        R, nb = rois5.shape[0], ph * pw
        K = C * nb
        pa = (0,) if pair else precision.patterns("gemm")[0]
        T, blk = len(pa), (K + 63) // 64 * 64
        planes = torch.empty((2 * R, T * blk), dtype=torch.bfloat16, device=feat.device)
        if blk != K:
            planes.zero_()                      # (never for C % 64 == 0 and 7 x 7: C*49 is then a multiple of 64)
        planes_cm = torch.empty((R, 2 * K), dtype=torch.bfloat16, device=feat.device) if pair else None
        # the fp32 pooled values are what the sampled-row views read -- except in the pair layout, where they are read from the
        # clean rows' cell-major planes (_RowViews, src_cm): 200 MB of this kernel's 700 not written (ODW_VIEWS_F32=1: as before)
        views_from_planes = pair and os.environ.get("ODW_VIEWS_F32") != "1"
        pooled32 = None if views_from_planes else torch.empty((R, C, ph, pw), dtype=torch.float32, device=feat.device)
        argmax = torch.empty((R, K), dtype=torch.int16, device=feat.device)
        lib = L.lib()
        ws_bytes = lib.odw_roi_pool_stack_nhwc_f32_workspace(R, B, C, H, W)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=feat.device)
        import ctypes
        pat = (ctypes.c_int * T)(*pa)
        with kernel_timer.region("roi_pool_stack_fwd_nhwc_f32",
                                 nbytes=float(B * C * H * W * 4 + 2 * R * T * K * 2 + (2 * R * K * 2 if pair else 0) + R * K * 2
                                              + (0 if pooled32 is None else R * K * 4)),
                                 alg=float(2 * R * K * 4 + B * C * H * W * 4)):
            L.check(lib.odw_roi_pool_stack_forward_nhwc_f32_cm(L.ptr(nhwc32), L.ptr(rois5), scale, B, C, H, W, R, L.ptr(keep),
                                                               L.ptr(keep_sum), ctypes.cast(pat, ctypes.c_void_p), T, L.ptr(planes),
                                                               planes.stride(0), blk, L.ptr(pooled32), L.ptr(argmax),
                                                               L.ptr(planes_cm), 2 * K if pair else 0, K if pair else 0,
                                                               L.ptr(ws), ws_bytes, L.stream()), "roi_pool_stack_forward_nhwc_f32")
        ctx.save_for_backward(rois5, keep, keep_sum, argmax)
        ctx.dims = (B, C, H, W, R, ph, pw)
        ctx.holder = holder
        ctx.set_materialize_grads(False)            # (or autograd hands backward 800 MB of zeros for the two)
        if pair:
            ctx.mark_non_differentiable(*[t for t in (planes, pooled32, planes_cm) if t is not None])
            return gemm.planes_handle(feat.device, 2 * R, K), planes, pooled32, planes_cm
        ctx.mark_non_differentiable(planes, pooled32)
        return gemm.planes_handle(feat.device, 2 * R, K), planes, pooled32

    @staticmethod
    def backward(ctx, dx, *unused):
        return _PoolStack.backward(ctx, dx) + (None,)


def _keep_sum(block):
    """Sum of a DropBlock keep mask: the value its kernel already produced, else a reduction."""
    total = getattr(block, "_odw_sum", None)
    return total if total is not None else block.sum()


class TwoFCROIFeatureExtractor(nn.Module):
    Linear = Linear
    fc_index = (1, 4)              # positions of the two Linear layers inside self.classifier

    def __init__(self, config):
        super().__init__()
        res = config.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        self.pooler = Pooler(output_size=(res, res), scales=config.MODEL.ROI_BOX_HEAD.POOLER_SCALES,
                             sampling_ratio=config.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO,
                             method=config.MODEL.ROI_BOX_HEAD.POOLER_METHOD)
        if config.DB.METHOD == "dropblock":
            self.dropblock = DropBlock2D(block_size=3, drop_prob=0.3)
        self.sim_drop = DropBlock2D(block_size=1, drop_prob=0.3)
        self.rand = None
        self._grad_holder = None
        self.sparse_clean = False
        self._clean_keys = None

    def init_fc(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    @property
    def fc6(self):
        return self.classifier[self.fc_index[0]]

    @property
    def fc7(self):
        return self.classifier[self.fc_index[1]]

    def _fc(self, x, segs6=None, segs7=None, grad_rows=None, row_ids=None, keep=None, pair=None):
        """Linear, ReLU, Dropout, Linear, ReLU, Dropout (vgg16.py:121-127).  With a counter-based `rand`
        the two dropouts are fused into the GEMM epilogues; `segs*` carry per-pass keys when
        several passes are stacked along the row dimension."""
        fc6, fc7 = self.fc6, self.fc7
        if not self.training:                   # inference: ReLU in the GEMM epilogue
            return fc7.fused(fc6.fused(x, relu=True), relu=True)
        if self.rand is None:                   # torch's generator instead of the counter-based streams
            x = F.dropout(fc6.fused(x, relu=True), 0.5, True)
            return F.dropout(fc7.fused(x, relu=True), 0.5, True)
        if segs6 is None:
            k6, k7 = self.rand.key(), self.rand.key()
            segs6, segs7 = [(0, k6[0], k6[1])], [(0, k7[0], k7[1])]
        if pair is not None:        # (planes_cm, planes_bwd, DropBlock keep mask, its sum): clean + DropBlock rows from one sweep
            x = fc6.pair(x, pair[0], pair[1], pair[2], pair[3], relu=True, drop_p=0.5, segs=segs6, grad_rows=grad_rows)
        else:
            x = fc6.fused(x, relu=True, drop_p=0.5, segs=segs6, grad_rows=grad_rows, row_ids=row_ids)
        if keep is not None:
            keep["h6"] = x.detach()
        return fc7.fused(x, relu=True, drop_p=0.5, segs=segs7, grad_rows=grad_rows, row_ids=row_ids)

    def forward_clean_and_aug(self, pooled):
        """The clean pass and the DropBlock pass of ROIWeakRegHead.forward (weak_head.py:107-112) as
        ONE stacked two-layer evaluation (M = 2P): random draws are numbered in the reference's
        order (clean fc6, clean fc7, DropBlock centres, aug fc6, aug fc7)."""
        P = pooled.shape[0]
        k1, k2 = self.rand.key(), self.rand.key()
        fused = (hasattr(self, "dropblock") and pooled.is_cuda and os.environ.get("ODW_NO_STACK_FUSE") != "1"
                 and pooled.dtype == torch.float32 and (pooled.shape[1] * pooled.shape[2] * pooled.shape[3]) % 64 == 0)
        if fused:
            # production path: the DropBlock view is never materialised in fp32 -- one kernel writes the stacked
            # bf16 GEMM operand, one kernel folds both halves of its gradient back (same draws, same arithmetic)
            block = self.dropblock.keep_mask(P, pooled.shape[2], pooled.shape[3], pooled.device, self.rand)
            k4, k5 = self.rand.key(), self.rand.key()
            if self._grad_holder is not None and self._grad_holder.pending:
                raise RuntimeError("the gradient of the previous step's sampled-row views was parked for the stacked "
                                   "fc6 node, whose backward never ran")
            self._grad_holder = _GradHolder()
            block_c, block_sum = block.contiguous(), _keep_sum(block)
            x = _StackCleanAug.apply(pooled, block_c, block_sum, self._grad_holder)
            pair = None
            if precision.get_precision() == "bf16x2f" and self.fc6.can_pair(pooled.shape[1], pooled.shape[2] * pooled.shape[3]):
                # the shared clean + DropBlock forward reads the CLEAN rows only (as cell-major planes); the stacked fp32
                # operand stays what the backward reads
                with torch.no_grad():
                    planes_cm = gemm.split_rows_cm(pooled.detach().reshape(P, -1), pooled.shape[1], pooled.shape[2] * pooled.shape[3])
                pair = (planes_cm, x.detach(), block_c.view(P, -1), block_sum)
            h = self._fc(x, segs6=[(0,) + k1, (P,) + k4], segs7=[(0,) + k2, (P,) + k5], pair=pair)
            return h[:P], h[P:]
        aug = self.forward_dropblock(pooled) if hasattr(self, "dropblock") else pooled
        k4, k5 = self.rand.key(), self.rand.key()
        x = torch.cat([pooled.reshape(P, -1), aug.reshape(P, -1)], dim=0)
        h = self._fc(x, segs6=[(0,) + k1, (P,) + k4], segs7=[(0,) + k2, (P,) + k5])
        return h[:P], h[P:]

    _grad_holder = None

    def sampled_row_views(self, pooled, groups, roi_index=None):
        """drop_pool and noise_pool of pooled[rows] for every (image, class) of the step, stacked for ONE
        fc6/fc7 evaluation (loss.py:292-305).  groups = [(row_base, rows int32 device tensor, k)]; `pooled` = the
        fp32 pooled tensor, or the stacked bf16 operand forward_pool_clean_and_aug returned (then roi_index = the
        int32 device list of the sampled ROIs, group after group).  Returns (x, segs6, segs7) with the random
        draws numbered in the reference's order (per group: drop mask, fc6, fc7 of the drop view; noise, fc6, fc7
        of the noise view), or None when the fused kernels do not apply."""
        planes = getattr(pooled, "_odw_planes", None)          # (bf16x2f) `pooled` is the handle of a planes operand
        stacked = pooled.dim() == 2 and (pooled.dtype == torch.bfloat16 or planes is not None)
        res = self.pooler.output_size
        S = res[0] * res[1]
        if not (self.rand is not None and pooled.is_cuda
                and (stacked or pooled.dtype == torch.float32) and self.sim_drop.block_size == 1 and S >= 4
                and (pooled.shape[1:].numel() % 64 == 0) and hasattr(self.rand, "key")):
            if stacked:
                raise RuntimeError("sampled_row_views: the stacked pooling path needs the fused kernels")
            return None
        holder = self._grad_holder
        if stacked:
            assert holder is not None and holder.kind == "extra" and roi_index is not None
            holder.roi_index = [roi_index]
        specs, segs6, segs7, row0 = [], [], [], 0
        for base, rows, k in groups:
            kd = self.rand.key()
            k6d, k7d = self.rand.key(), self.rand.key()
            kn = self.rand.key()
            k6n, k7n = self.rand.key(), self.rand.key()
            specs.append((base, rows, k, kd, kn))
            segs6 += [(row0,) + k6d, (row0 + k,) + k6n]
            segs7 += [(row0,) + k7d, (row0 + k,) + k7n]
            row0 += 2 * k
        pooled32 = pooled._odw_pooled32 if planes is not None else None
        if planes is not None and pooled32 is None:         # the views straight as the operand of fc6 (its planes)
            x, hi, cm = _RowViews.apply(pooled, specs, holder, float(self.sim_drop.drop_prob), S, None, pooled._odw_planes_cm)
            x._odw_planes, x._odw_planes_cm = hi, cm
            return x, segs6, segs7
        x = _RowViews.apply(pooled, specs, holder, float(self.sim_drop.drop_prob), S, pooled32)
        return x, segs6, segs7

    def can_pool_stack(self, features):
        """ROI pooling may write the stacked bf16 operand directly (csrc/roi_pool.hip: roi_pool_stack_*; its 32-bit
        (value, position) keys hold bf16 values: a split precision mode pools in fp32 with the operator form)."""
        if not (self.rand is not None and hasattr(self, "dropblock")
                and self.training and len(features) == 1 and os.environ.get("ODW_NO_POOL_STACK") != "1"):
            return False
        f = features[0]
        if precision.split_mode():
            # "bf16x2f": the fp32 NHWC map pooled straight into the bf16 planes of the operand (_PoolStackPlanes); the
            # modes whose backward is split too keep the operator form (their weight gradient wants the fp32 operand)
            if precision.bwd_split() or getattr(f, "_odw_nhwc_f32", None) is None or f.shape[1] % 64 != 0:
                return False
        pool = self.pooler.poolers[0]
        res = self.pooler.output_size
        return (type(pool).__name__ == "ROIPool" and f.is_cuda and f.dtype == torch.float32 and f.dim() == 4
                and f.shape[2] * f.shape[3] < 65535 and 4 * f.shape[2] * f.shape[3] * 4 <= 160 * 1024
                and (f.shape[1] * res[0] * res[1]) % 64 == 0 and res[0] * res[1] <= 128)

    def forward_pool_clean_and_aug(self, features, proposals):
        """forward_pooler + forward_clean_and_aug with NO pooled tensor in between: the pooling kernel writes the
        stacked (2P x C*7*7) bf16 operand (clean rows, DropBlock rows) and a 16-bit argmax.  Same random draws in
        the same order.  Returns (clean_feats, aug_feats, stacked operand) -- the operand stands in for
        `clean_pooled` in the loss (sampled_row_views reads its first P rows)."""
        feat = features[0]
        rois5 = self.pooler.convert_to_roi_format(proposals)
        P = rois5.shape[0]
        res = self.pooler.output_size
        k1, k2 = self.rand.key(), self.rand.key()
        block = self.dropblock.keep_mask(P, res[0], res[1], feat.device, self.rand)
        k4, k5 = self.rand.key(), self.rand.key()
        if self._grad_holder is not None and self._grad_holder.pending:
            raise RuntimeError("the gradient of the previous step's sampled-row views was never folded")
        self._grad_holder = _GradHolder("extra")
        self._grad_holder.grad_out = getattr(feat, "_odw_grad_out", None)
        nhwc = getattr(feat, "_odw_nhwc", None) if os.environ.get("ODW_POOL_NHWC") != "0" else None
        pair = None
        if precision.split_mode():
            # the shared clean + DropBlock fc6 forward (gemm.pair_linear); fc6 then keeps its weight as cell-major planes only
            # (ODW_NO_PAIR=1: the stacked pass of rounds 1-3 over channel-major planes, for comparison)
            use_pair = precision.get_precision() == "bf16x2f" and self.fc6.can_pair(feat.shape[1], res[0] * res[1])
            if not use_pair and precision.get_precision() == "bf16x2f" and self.fc6._get_shadow().cm is not None:
                raise RuntimeError("forward_pool_clean_and_aug: fc6 keeps cell-major planes (no channel-major forward copy) "
                                   "but the shared clean + DropBlock forward does not apply")
            block_c, block_sum = block.contiguous(), _keep_sum(block)
            out = _PoolStackPlanes.apply(feat, feat._odw_nhwc_f32, rois5, block_c, block_sum, self._grad_holder,
                                         float(self.pooler.poolers[0].spatial_scale), res[0], res[1], use_pair)
            x, planes, pooled32 = out[0], out[1], out[2]
            x._odw_planes, x._odw_pooled32 = planes, pooled32
            if use_pair:
                x._odw_planes_cm = out[3]       # (rows [0, P) only: the clean half)
                pair = (out[3], planes, block_c.view(P, -1), block_sum)
                if self._grad_holder is not None and os.environ.get("ODW_ABSMAX_PREPASS") != "1":
                    # fc6's input-gradient GEMM leaves max |dx| with the holder: the pooling backward's scale without its
                    # 200 MB pre-pass (gemm._backward_single_plane, csrc/gemm_bf16.hip: epi_absmax_commit)
                    x._odw_absmax_holder = self._grad_holder
        else:
            x = _PoolStack.apply(feat, rois5, block.contiguous(), _keep_sum(block), self._grad_holder,
                                 float(self.pooler.poolers[0].spatial_scale), res[0], res[1], nhwc)
        # The clean half feeds only Sim_Net, and the contrastive loss touches a few hundred of its P rows: the stacked
        # evaluation takes part in backward with its DropBlock half only (grad_rows); the clean rows the loss ends
        # up using are re-evaluated by recompute_clean_rows with their original dropout draws (row_ids).
        self.sparse_clean = os.environ.get("ODW_NO_SPARSE") != "1"
        self._clean_keys = (k1, k2)
        keep = {} if self.sparse_clean else None
        h = self._fc(x, segs6=[(0,) + k1, (P,) + k4], segs7=[(0,) + k2, (P,) + k5],
                     grad_rows=(P, 2 * P, False) if self.sparse_clean else None, keep=keep, pair=pair)   # consumers skip the clean half:
        self._grad_holder.clean_rows = 0 if self.sparse_clean else P                       # fc6's slices it, pooling below
        # the clean half's fc6 / fc7 outputs stay around: the rows the contrastive loss differentiates are re-attached
        # to the graph from them (reuse_clean_rows) instead of being evaluated a second time
        self._clean_acts = (keep["h6"], h.detach()) if self.sparse_clean else None
        return (h[:P].detach() if self.sparse_clean else h[:P]), h[P:], x

    def recompute_clean_rows(self, stacked, rows, first_entry):
        """fc6/fc7 of the clean rows `rows` (int32 device list, ascending) of the stacked operand, WITH autograd and
        with the dropout draws those rows had in the stacked pass; their input gradient is parked as extra rows
        [first_entry, first_entry + len(rows)) of the pooling node's side buffer."""
        k1, k2 = self._clean_keys
        planes, planes_cm = getattr(stacked, "_odw_planes", None), getattr(stacked, "_odw_planes_cm", None)
        if planes is not None and planes_cm is not None:
            x, picked, picked_cm = _RowGather.apply(stacked, rows, self._grad_holder, first_entry, planes, planes_cm)
            x._odw_planes, x._odw_planes_cm = picked, picked_cm
        elif planes is not None:
            x, picked = _RowGather.apply(stacked, rows, self._grad_holder, first_entry, planes)
            x._odw_planes = picked
        else:
            x = _RowGather.apply(stacked, rows, self._grad_holder, first_entry)
        return self._fc(x, segs6=[(0,) + k1], segs7=[(0,) + k2], row_ids=rows)

    def reuse_clean_rows(self, stacked, rows, first_entry):
        """The same rows as recompute_clean_rows WITHOUT evaluating them again: their fc6 / fc7 outputs were computed by
        the stacked pass (same inputs, same dropout draws); they are gathered and re-attached to the graph
        (Linear.reuse), so that backward runs over them as usual and their input gradient is parked for the pooling
        node.  The result carries `_odw_reuse_rows` so that Sim_Net does the same with ITS stacked-pass outputs."""
        h6, h7 = self._clean_acts
        planes = getattr(stacked, "_odw_planes", None)
        if planes is not None:
            x, picked = _RowGather.apply(stacked, rows, self._grad_holder, first_entry, planes)
            x._odw_planes = picked
        else:
            x = _RowGather.apply(stacked, rows, self._grad_holder, first_entry)
        a6 = self.fc6.reuse(x, h6, rows, relu=True, drop_p=0.5)
        a7 = self.fc7.reuse(a6, h7, rows, relu=True, drop_p=0.5)
        a7._odw_reuse_rows = rows
        return a7

    def forward(self, x, proposals):
        pooled = self.pooler(x, proposals)
        return self._fc(pooled.reshape(pooled.shape[0], -1)), pooled

    def forward_pooler(self, x, proposals):
        return self.pooler(x, proposals)

    def forward_neck(self, x):
        return self._fc(x.reshape(x.shape[0], -1))

    def forward_dropblock(self, pooled_feats, proposals=None):
        return self.dropblock(pooled_feats, self.rand)

    def drop_pool(self, pooled_feats):
        return self.sim_drop(pooled_feats, self.rand)

    def noise_pool(self, pooled_feats):
        if self.rand is not None:
            return self.rand.noise_mul(pooled_feats)
        noise = torch.randn_like(pooled_feats)
        return noise * pooled_feats + pooled_feats
