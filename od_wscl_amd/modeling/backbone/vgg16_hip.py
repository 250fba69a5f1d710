"""VGG16-OICR backbone on the gfx950 kernels: NHWC bf16 activations, every 3x3 convolution an
implicit GEMM on the MFMA tile (csrc/gemm_bf16.hip:conv3x3_glds_kernel) with bias+ReLU fused,
forward AND backward as one autograd node (the reference: 13 cuDNN convs + 12 ReLUs + 3 pools as
separate autograd nodes, modeling/backbone/vgg16.py:34-36,58-83).

The parameters stay the nn.Conv2d weights of `VGG_Base.features` (same names, same layout for
checkpoints and the optimiser); packed bf16 copies are refreshed from them at the start of each
forward.  Frozen layers (FREEZE_CONV_BODY_AT=2 -> conv1_x, conv2_x) run forward only, and no input
gradient is computed below the first trainable convolution."""
import collections
import logging
import os

import torch
from torch import nn

from ... import _lib as L
from ... import gemm
from ... import precision as P
from ...utils.kernel_timer import kernel_timer


def _r64(n):
    return (n + 63) // 64 * 64


class _Layer(object):
    __slots__ = ("conv", "cin", "cp", "cout", "dil", "relu", "pool", "trainable", "wk", "wd", "mode", "packed_version", "p2")


def _layers_of(features):
    mods = list(features)
    out = []
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            l = _Layer()
            l.conv, l.cin, l.cout, l.dil = m, m.in_channels, m.out_channels, m.dilation[0]
            assert m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding[0] == m.dilation[0]
            l.cp = max(8, 1 << (l.cin - 1).bit_length())
            l.relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            j = i + (2 if l.relu else 1)
            l.pool = j < len(mods) and isinstance(mods[j], nn.MaxPool2d)
            l.trainable = m.weight.requires_grad
            l.wk = l.wd = l.mode = l.packed_version = l.p2 = None
            out.append(l)
        i += 1
    return out


def _conv3x3(lib, x, m, h, w, c, dil, mirror, wk, n, y, bias, relu, mask, ldmask, zero_page, st, flops, alg=None):
    """One implicit-GEMM convolution (y bf16 or fp32); the launcher's split-K workspace (deep layers) comes from
    torch's allocator.  flops = MFMA work issued (plane products counted), alg = the one fp32 product they stand for."""
    ws_bytes = lib.odw_conv3x3_workspace_hw(m, h, w, c, n, dil)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=y.device) if ws_bytes else None
    out_bf16 = y.dtype == torch.bfloat16
    sym = "conv3x3 split-K+reduce" if ws_bytes else ("conv3x3<%s>" % ("bf16" if out_bf16 else "f32"))
    with kernel_timer.region(sym, flops=flops, alg=alg, shape="m=%d,hw=%dx%d,c=%d,n=%d,dil=%d" % (m, h, w, c, n, dil)):
        L.check(lib.odw_conv3x3_nhwc_bf16_ws(L.ptr(x), m, h, w, c, dil, mirror, L.ptr(wk), wk.stride(0), n, L.ptr(y), n,
                                             1 if out_bf16 else 0, L.ptr(bias), 1 if relu else 0, L.ptr(mask), ldmask,
                                             L.ptr(zero_page), L.ptr(ws), ws_bytes, st), "conv3x3")


def planes2_layer(l):
    """True when the "bf16x2f" forward of this layer runs on the two stored planes [hi | mid] of its input
    (csrc/gemm_bf16.hip: conv3x3_halo2_kernel) instead of three passes over the blocks [hi | hi | mid]."""
    return l.cp == l.cin and l.cp % 32 == 0 and l.cout % 64 == 0 and l.dil in (1, 2)


def _wgrad_tn(l):
    """The layer's weight gradient runs on the K-major (TN / halo) kernels, which read dZ and the layer input as they are."""
    return l.cp >= 128 and l.cout % 8 == 0 and os.environ.get("ODW_CONV_WGRAD_TN") != "0"


def _wgrad_reads_strided(l):
    """ONE predicate for "the backward reads the hi plane of the forward's operand in place" (a strided view of the plane
    buffer, which then stays alive through the backward), used by the forward when it saves the operand AND implied by the
    backward's kernel choice: the K-major kernels (_wgrad_tn) on shapes their halo form takes (64-channel granules, dilation
    1 / 2).  Everything else gets a dense copy and the plane buffer is released after the forward."""
    return _wgrad_tn(l) and l.cout % 64 == 0 and l.cp % 64 == 0 and l.dil in (1, 2)


def planes2_body(net):
    """The whole body takes the two-plane path: a direct stem that writes planes, every later layer eligible
    (ODW_CONV_PLANES2=0: the three-pass form of rounds 3-4, for comparison runs)."""
    l0 = net.layers[0]
    return (os.environ.get("ODW_CONV_PLANES2") != "0" and len(net.layers) > 1 and not l0.trainable and l0.cin == 3 and l0.dil == 1
            and l0.relu and not l0.pool and l0.cout % 32 == 0 and os.environ.get("ODW_NO_STEM") != "1"
            and all(planes2_layer(l) for l in net.layers[1:]))


def _conv3x3_planes2(lib, xs, m, h, w, l, y, y_planes, zero_page, st):
    """One forward convolution over the two-plane operand xs (m, >= 2 cp) -> y fp32 (m, cout) [y_planes 0], planes (m, 2 cout)
    [1], or -- 2 -- the planes of the 2 x 2 max-pooled result (m / 4, 2 cout): pool and split in the epilogue."""
    ws_bytes = lib.odw_conv3x3_planes2_workspace(m, h, w, l.cp, l.cout)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=y.device) if ws_bytes else None
    if y_planes == 2:
        ws_bytes, ws = 0, None              # (the pooled epilogue takes whole sums: the launcher does not slice K)
    sym = "conv3x3_planes2<%s>%s" % (("f32", "planes", "pool+planes")[int(y_planes)], " split-K+reduce" if ws_bytes else "")
    issued = 2.0 * m * l.cout * 9 * l.cin * 3
    with kernel_timer.region(sym, flops=issued, alg=issued / 3.0, shape="m=%d,hw=%dx%d,c=%d,n=%d,dil=%d" % (m, h, w, l.cp, l.cout, l.dil)):
        L.check(lib.odw_conv3x3_planes2_ws(L.ptr(xs), xs.stride(0), m, h, w, l.cp, l.dil, L.ptr(l.wk), l.wk.stride(0), l.cout,
                                           L.ptr(y), y.stride(0), int(y_planes), L.ptr(l.conv.bias), 1 if l.relu else 0,
                                           L.ptr(zero_page), L.ptr(ws), ws_bytes, st), "conv3x3_planes2")


def conv_wgrad(lib, dzt, colt, cout, cin, cp, k, dw, st, accumulate=0):
    """dw (Cout, Cin, 3, 3) fp32 = dZ^T im2col(X) over k reduction columns (pixels; plane products in a split mode) as
    ONE call: split-K partial products, then one pass that reduces the slices and unpacks into torch's layout."""
    ws_bytes = lib.odw_conv_wgrad_workspace(cout, cp, k, dzt.stride(0), colt.stride(0))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dzt.device)
    with kernel_timer.region("conv wgrad split-K+reduce", flops=2.0 * cout * 9 * cp * k):
        L.check(lib.odw_conv_wgrad_nt(L.ptr(dzt), dzt.stride(0), L.ptr(colt), colt.stride(0), cout, cin, cp, k, L.ptr(dw),
                                      accumulate, L.ptr(ws), ws_bytes, st), "conv_wgrad_nt")


class _VGGFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, net, *params):
        lib = L.lib()
        B, C, H, W = images.shape
        dev = images.device
        st = L.stream()
        l0 = net.layers[0]
        # a frozen 3-channel first layer runs as a direct kernel on the fp32 NCHW image (K = 27 wastes the MFMA tile)
        direct0 = (not l0.trainable and l0.cin == 3 and l0.dil == 1 and l0.relu and l0.cout % 8 == 0
                   and os.environ.get("ODW_NO_STEM") != "1")
        x = None
        if not direct0:
            x = torch.empty((B * H * W, 8), dtype=torch.bfloat16, device=dev)
            L.check(lib.odw_nchw_f32_to_nhwc_bf16(L.ptr(images.contiguous()), B, H * W, C, 8, L.ptr(x), st), "nchw_to_nhwc")
        saved = []          # per layer: (input activation, pre-pool activation or None, H, W)
        h, w = H, W
        for li, l in enumerate(net.layers):
            m = B * h * w
            y = torch.empty((m, l.cout), dtype=torch.bfloat16, device=dev)
            if li == 0 and direct0:
                L.check(lib.odw_stem_conv3x3_bias_relu(L.ptr(images.contiguous()), L.ptr(l.conv.weight.detach()),
                                                       L.ptr(l.conv.bias.detach()), B, H, W, l.cout, L.ptr(y), st), "stem_conv3x3")
            else:
                _conv3x3(lib, x, m, h, w, l.cp, l.dil, 0, l.wk, l.cout, y, l.conv.bias, l.relu, None, 0, net.zero_page, st,
                         2.0 * m * l.cout * 9 * l.cin)
            pre = None
            if l.pool:
                pre = y
                p = torch.empty((B * (h // 2) * (w // 2), l.cout), dtype=torch.bfloat16, device=dev)
                L.check(lib.odw_maxpool2x2_nhwc_bf16(L.ptr(y), B, h, w, l.cout, L.ptr(p), st), "maxpool")
                y = p
            saved.append((x if l.trainable else None, pre if l.trainable else None, h, w))
            if l.pool:
                h, w = h // 2, w // 2
            x = y
        feat = torch.empty((B, net.layers[-1].cout, h, w), dtype=torch.float32, device=dev)
        L.check(lib.odw_nhwc_bf16_to_nchw_f32(L.ptr(x), B, h * w, net.layers[-1].cout, L.ptr(feat), st), "nhwc_to_nchw")
        ctx.net, ctx.saved_acts, ctx.batch = net, saved, B
        net.last_nhwc = x           # the NHWC bf16 map itself: the fused ROI pooling reads it directly
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        return _backward_single_plane(ctx, dfeat)


def backward_segments(net):
    """The trainable layers in backward order, cut into net.bwd_segments runs of consecutive layers [(hi, lo), ...]
    (inclusive): data-parallel runs hand each run's weight gradients to the exchange as it completes
    (net.on_segment_done, engine.build_training_step) instead of all of them after the whole backward."""
    first = min(i for i, l in enumerate(net.layers) if l.trainable)
    layers = list(range(len(net.layers) - 1, first - 1, -1))
    n = max(1, min(int(getattr(net, "bwd_segments", 1) or 1), len(layers)))
    per = -(-len(layers) // n)
    return [(layers[i], layers[min(i + per, len(layers)) - 1]) for i in range(0, len(layers), per)]


def _backward_single_plane(ctx, dfeat, seg=None):
    """Backward of the body with one bf16 plane per operand ("bf16", and "bf16x2f" after its split-precision
    forward): saved layer inputs are NHWC bf16; a pooled layer's saved pre-pool activation is bf16 or ("bf16x2f") the
    fp32 tensor of the forward, whose first maximum routes the gradient.
    seg = k: only the k-th run of backward_segments(net) (graph capture: one graph per run; the gradient that travels
    between runs is kept on ctx); seg = None: everything, announcing each completed run to net.on_segment_done."""
    lib = L.lib()
    net, saved, B = ctx.net, ctx.saved_acts, ctx.batch
    st = L.stream()
    dev = dfeat.device
    segs = backward_segments(net)
    first = segs[-1][1]
    if seg is None or seg == 0:
        _, C, h, w = dfeat.shape
        dz = torch.empty((B * h * w, C), dtype=torch.bfloat16, device=dev)
        L.check(lib.odw_nchw_f32_to_nhwc_bf16(L.ptr(dfeat.contiguous()), B, h * w, C, C, L.ptr(dz), st), "nchw_to_nhwc")
    else:
        dz = ctx._bw_dz
    acc = 1 if getattr(net, "accumulate", False) else 0      # SOLVER.ITER_SIZE: later micro-steps add to the gradients
    seg_end = {lo: k for k, (hi, lo) in enumerate(segs)}
    hi_li, lo_li = (segs[0][0], segs[-1][1]) if seg is None else segs[seg]
    done_cb = getattr(net, "on_segment_done", None) if seg is None else None
    for li in range(hi_li, lo_li - 1, -1):
        l = net.layers[li]
        x_in, pre, h, w = saved[li]
        if l.pool:       # dz arrives at the pooled resolution: route through the pool (+ ReLU mask of `pre`)
            d_pre = torch.empty((B * h * w, l.cout), dtype=torch.bfloat16, device=dev)
            pool_bwd = lib.odw_maxpool2x2_nhwc_f32x_bf16_bwd if pre.dtype == torch.float32 else lib.odw_maxpool2x2_nhwc_bf16_bwd
            L.check(pool_bwd(L.ptr(pre), L.ptr(dz), B, h, w, l.cout, L.ptr(d_pre), st), "maxpool_bwd")
            dz = d_pre
        m = B * h * w
        m64 = _r64(m)
        if getattr(net, "debug", None) is not None:      # tests: gradient w.r.t. this layer's pre-activation
            net.debug[li] = dz.float().reshape(B, h, w, l.cout).permute(0, 3, 1, 2).clone()
        # ---- bias + weight gradient
        conv = l.conv
        if conv.bias.grad is None:
            conv.bias.grad = torch.zeros_like(conv.bias)
        if conv.weight.grad is None:
            conv.weight.grad = torch.empty_like(conv.weight)
        if _wgrad_tn(l):
            # dZ and the layer input as they are (NHWC rows): K-major operands, transposed fragment reads; the bias
            # gradient (column sums of dZ) comes out of the same launch
            ws_bytes = lib.odw_conv_wgrad_tn_bias_workspace(l.cout, l.cp, m)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            with kernel_timer.region("conv wgrad TN split-K+reduce", flops=2.0 * m * l.cout * 9 * l.cp):
                L.check(lib.odw_conv_wgrad_tn_bias(L.ptr(dz), l.cout, L.ptr(x_in), x_in.stride(0), m, h, w, l.cp, l.dil, l.cout, l.cin,
                                                   L.ptr(conv.weight.grad), L.ptr(conv.bias.grad), acc, L.ptr(net.zero_page),
                                                   L.ptr(ws), ws_bytes, st), "conv_wgrad_tn_bias")
        else:
            dzc = torch.empty((m, _r64(l.cout)), dtype=torch.bfloat16, device=dev)
            dzt = torch.empty((l.cout, m64), dtype=torch.bfloat16, device=dev)
            L.check(lib.odw_linear_bwd_prep(L.ptr(dz), 0, l.cout, None, 0, m, l.cout, 1.0, L.ptr(dzc), dzc.stride(0),
                                            L.ptr(dzt), m64, L.ptr(conv.bias.grad), st), "conv bias grad")
            colt = torch.empty((9 * l.cp, m64), dtype=torch.bfloat16, device=dev)
            L.check(lib.odw_im2col_t_bf16(L.ptr(x_in.contiguous()), m, h, w, l.cp, l.dil, L.ptr(colt), m64, st), "im2col_t")
            conv_wgrad(lib, dzt, colt, l.cout, l.cin, l.cp, m, conv.weight.grad, st, acc)
        # ---- input gradient (masked by the ReLU of the producing layer unless that layer was pooled:
        #      then the pool backward of the previous iteration applies the mask)
        if li > first:
            prev = net.layers[li - 1]
            dx = torch.empty((m, l.cin), dtype=torch.bfloat16, device=dev)
            mask = x_in if (prev.relu and not prev.pool) else None
            _conv3x3(lib, dz, m, h, w, l.cout, l.dil, 1, l.wd, l.cin, dx, None, False, mask,
                     mask.stride(0) if mask is not None else 0, net.zero_page, st, 2.0 * m * l.cout * 9 * l.cin)
            dz = dx
        if done_cb is not None and li in seg_end and len(segs) > 1:
            done_cb(seg_end[li])
    ctx._bw_dz = dz if seg is not None else None
    return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _VGGSplitFn(torch.autograd.Function):
    """The same backbone node in a split precision mode (precision.py, "bf16x3" = fp32-grade): activations are fp32
    NHWC between kernels; each convolution's input is laid out as bf16 planes along the channel axis
    (csrc/split.hip: [hi hi hi mid mid lo 0 0] x Cp channels against weights packed [hi mid lo hi mid hi 0 0]) and
    the unchanged implicit-GEMM kernel runs over 8*Cp channels with an fp32 epilogue.  Weight gradient: the planes
    of dZ^T and of the transposed im2col as column blocks of the split-K GEMM's operands."""

    @staticmethod
    def forward(ctx, images, net, *params):
        lib = L.lib()
        B, C, H, W = images.shape
        dev = images.device
        st = L.stream()
        pa, _ = P.patterns("conv")
        T = len(pa)
        cp0 = net.layers[0].cp
        x = torch.empty((B * H * W, cp0), dtype=torch.float32, device=dev)
        L.check(lib.odw_nchw_f32_to_nhwc_f32(L.ptr(images.contiguous()), B, H * W, C, cp0, L.ptr(x), st), "nchw_to_nhwc_f32")
        saved = []
        h, w = H, W
        for l in net.layers:
            m = B * h * w
            xs = P.split_rows(x, pa, l.cp)
            y = torch.empty((m, l.cout), dtype=torch.float32, device=dev)
            _conv3x3(lib, xs, m, h, w, T * l.cp, l.dil, 0, l.wk, l.cout, y, l.conv.bias, l.relu, None, 0, net.zero_page, st,
                     2.0 * m * l.cout * 9 * l.cin * T, alg=2.0 * m * l.cout * 9 * l.cin)
            del xs
            pre = None
            if l.pool:
                pre = y
                p = torch.empty((B * (h // 2) * (w // 2), l.cout), dtype=torch.float32, device=dev)
                L.check(lib.odw_maxpool2x2_nhwc_f32(L.ptr(y), B, h, w, l.cout, L.ptr(p), st), "maxpool_f32")
                y = p
            saved.append((x if l.trainable else None, pre if l.trainable else None, h, w))
            if l.pool:
                h, w = h // 2, w // 2
            x = y
        cl = net.layers[-1].cout
        feat = torch.empty((B, cl, h, w), dtype=torch.float32, device=dev)
        L.check(lib.odw_nhwc_f32_to_nchw_f32(L.ptr(x), B, h * w, cl, cl, L.ptr(feat), st), "nhwc_to_nchw_f32")
        ctx.net, ctx.saved_acts, ctx.batch = net, saved, B
        net.last_nhwc = None
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        lib = L.lib()
        net, saved, B = ctx.net, ctx.saved_acts, ctx.batch
        st = L.stream()
        dev = dfeat.device
        pa, _ = P.patterns("conv")
        ga, gb = P.patterns("gemm")
        T, Tg = len(pa), len(ga)
        _, C, h, w = dfeat.shape
        dz = torch.empty((B * h * w, C), dtype=torch.float32, device=dev)
        L.check(lib.odw_nchw_f32_to_nhwc_f32(L.ptr(dfeat.contiguous().float()), B, h * w, C, C, L.ptr(dz), st), "nchw_to_nhwc_f32")
        first = min(i for i, l in enumerate(net.layers) if l.trainable)
        for li in range(len(net.layers) - 1, first - 1, -1):
            l = net.layers[li]
            x_in, pre, h, w = saved[li]
            if l.pool:
                d_pre = torch.empty((B * h * w, l.cout), dtype=torch.float32, device=dev)
                L.check(lib.odw_maxpool2x2_nhwc_f32_bwd(L.ptr(pre), L.ptr(dz), B, h, w, l.cout, L.ptr(d_pre), st), "maxpool_f32_bwd")
                dz = d_pre
            m = B * h * w
            m64 = _r64(m)
            if getattr(net, "debug", None) is not None:
                net.debug[li] = dz.reshape(B, h, w, l.cout).permute(0, 3, 1, 2).clone()
            conv = l.conv
            if conv.bias.grad is None:
                conv.bias.grad = torch.zeros_like(conv.bias)
            if conv.weight.grad is None:
                conv.weight.grad = torch.empty_like(conv.weight)
            # bias gradient = column sums of dZ (single writer per entry; dZ passes through unchanged)
            P.bwd_mask(dz, None, 1.0, conv.bias.grad, out=dz)
            # weight gradient: dWk[co][tap*Cp+ci] = sum over pixels and plane products
            dzt = P.split_cols(dz, ga, m64)                                   # (Cout, Tg*m64)
            planes = [P.split_rows(x_in, (p,), l.cp) for p in (0, 1, 2)]      # hi / mid / lo of the layer input, NHWC bf16
            colt = torch.empty((9 * l.cp, Tg * m64), dtype=torch.bfloat16, device=dev)
            for t, pl in enumerate(gb):
                L.check(lib.odw_im2col_t_bf16_part(L.ptr(planes[pl]), m, h, w, l.cp, l.dil, L.ptr(colt[:, t * m64:]),
                                                   Tg * m64, m64, st), "im2col_t")
            conv_wgrad(lib, dzt, colt, l.cout, l.cin, l.cp, Tg * m64, conv.weight.grad, st,
                       1 if getattr(net, "accumulate", False) else 0)
            del dzt, colt
            if li > first:
                prev = net.layers[li - 1]
                dzs = P.split_rows(dz, pa, l.cout)
                dx = torch.empty((m, l.cin), dtype=torch.float32, device=dev)
                mask = planes[0] if (prev.relu and not prev.pool) else None     # hi plane: zero exactly where x_in is
                _conv3x3(lib, dzs, m, h, w, T * l.cout, l.dil, 1, l.wd, l.cin, dx, None, False, mask,
                         l.cp if mask is not None else 0, net.zero_page, st, 2.0 * m * l.cout * 9 * l.cin * T,
                         alg=2.0 * m * l.cout * 9 * l.cin)
                dz = dx
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _VGGMixedFn(torch.autograd.Function):
    """Precision mode "bf16x2f": the forward of _VGGSplitFn (fp32 NHWC activations, every convolution over the bf16
    planes of its input -- the feature map the ROI pooling, the losses and every selection depend on is fp32-grade),
    the backward of _VGGFn (dZ, layer inputs and weights each as one bf16 plane).  What is saved per trainable layer is
    the hi plane of its input (NHWC bf16: the weight-gradient operand and the ReLU mask) and, for a pooled layer, the
    fp32 pre-pool activation."""

    @staticmethod
    def forward(ctx, images, net, *params):
        lib = L.lib()
        B, C, H, W = images.shape
        dev = images.device
        st = L.stream()
        import ctypes
        l0 = net.layers[0]
        # a frozen 3-channel first layer runs as the direct fp32 kernel of the bf16 mode (an fmaf chain per output:
        # fp32-grade as it is) and writes the PLANES the second layer reads -- no NHWC copy of the image, no K = 27
        # MFMA pass, no fp32 activation + split pass
        direct0 = (not l0.trainable and l0.cin == 3 and l0.dil == 1 and l0.relu and not l0.pool and l0.cout % 8 == 0
                   and len(net.layers) > 1 and os.environ.get("ODW_NO_STEM") != "1")
        if planes2_body(net):
            return _VGGMixedFn._forward_planes2(ctx, images, net)
        x = None
        if not direct0:
            cp0 = l0.cp
            x = torch.empty((B * H * W, cp0), dtype=torch.float32, device=dev)
            L.check(lib.odw_nchw_f32_to_nhwc_f32(L.ptr(images.contiguous()), B, H * W, C, cp0, L.ptr(x), st), "nchw_to_nhwc_f32")
        saved = []
        h, w = H, W
        xs_ready = None                     # the next layer's plane operand, when its producer wrote it itself
        for li, l in enumerate(net.layers):
            m = B * h * w
            if li == 0 and direct0:
                nxt = net.layers[1]
                pn, _ = P.conv_patterns(nxt.cp)
                xs_ready = torch.empty((m, len(pn) * nxt.cp), dtype=torch.bfloat16, device=dev)
                if nxt.cp != l.cout:
                    xs_ready.zero_()
                cpat = (ctypes.c_int * len(pn))(*pn)
                L.check(lib.odw_stem_conv3x3_bias_relu_planes(L.ptr(images.contiguous()), L.ptr(l.conv.weight.detach()),
                                                              L.ptr(l.conv.bias.detach()), B, H, W, l.cout,
                                                              ctypes.cast(cpat, ctypes.c_void_p), len(pn), L.ptr(xs_ready),
                                                              xs_ready.stride(0), nxt.cp, st), "stem_conv3x3_planes")
                saved.append((None, None, h, w))
                continue
            pa, _ = P.conv_patterns(l.cp)
            T = len(pa)
            if xs_ready is not None:
                xs, xs_ready = xs_ready, None
                assert not l.trainable
                x16 = None
            else:
                xs = P.split_rows(x, pa, l.cp)
                # the backward's operand (weight gradient, ReLU mask) is the hi plane: the FIRST block of the planes, read
                # in place with row stride T * cp (was: a second split pass per layer writing a contiguous copy)
                x16 = xs[:, :l.cp] if (l.trainable and pa[0] == 0 and _wgrad_reads_strided(l)) else (
                    P.split_rows(x, (0,), l.cp) if l.trainable else None)
            y = torch.empty((m, l.cout), dtype=torch.float32, device=dev)
            _conv3x3(lib, xs, m, h, w, T * l.cp, l.dil, 0, l.wk, l.cout, y, l.conv.bias, l.relu, None, 0, net.zero_page, st,
                     2.0 * m * l.cout * 9 * l.cin * T, alg=2.0 * m * l.cout * 9 * l.cin)
            del xs
            pre = None
            if l.pool:
                pre = y
                p = torch.empty((B * (h // 2) * (w // 2), l.cout), dtype=torch.float32, device=dev)
                L.check(lib.odw_maxpool2x2_nhwc_f32(L.ptr(y), B, h, w, l.cout, L.ptr(p), st), "maxpool_f32")
                y = p
            saved.append((x16, pre if l.trainable else None, h, w))
            if l.pool:
                h, w = h // 2, w // 2
            x = y
        cl = net.layers[-1].cout
        feat = torch.empty((B, cl, h, w), dtype=torch.float32, device=dev)
        L.check(lib.odw_nhwc_f32_to_nchw_f32(L.ptr(x), B, h * w, cl, cl, L.ptr(feat), st), "nhwc_to_nchw_f32")
        ctx.net, ctx.saved_acts, ctx.batch = net, saved, B
        net.last_nhwc = None
        net.last_nhwc_f32 = x       # the fp32 NHWC map (read by the fused pooling of this mode)
        return feat

    @staticmethod
    def _forward_planes2(ctx, images, net):
        """Round 5: every activation between two convolutions exists ONLY as the next layer's operand -- the two bf16 planes
        [hi (C) | mid (C)] per pixel, written by the producing kernel itself: the stem, the convolution epilogue
        (conv3x3_halo2_kernel, OUTM = 1), or the pooling kernel behind a pooled layer.  No fp32 activation and no
        split_rows pass per layer (12 launches and 6 + 4 bytes per element moved in rounds 3-4); fp32 only where something
        else reads it: a pooled layer's pre-pool activation (its first maximum routes the gradient) and the feature map.
        The hi plane -- the first C columns of a row -- is the backward's operand in place (weight gradient, ReLU mask)."""
        import ctypes
        lib = L.lib()
        B, C, H, W = images.shape
        dev = images.device
        st = L.stream()
        l0, l1 = net.layers[0], net.layers[1]
        m = B * H * W
        xs = torch.empty((m, 2 * l1.cp), dtype=torch.bfloat16, device=dev)
        cpat = (ctypes.c_int * 2)(0, 1)
        L.check(lib.odw_stem_conv3x3_bias_relu_planes(L.ptr(images.contiguous()), L.ptr(l0.conv.weight.detach()),
                                                      L.ptr(l0.conv.bias.detach()), B, H, W, l0.cout,
                                                      ctypes.cast(cpat, ctypes.c_void_p), 2, L.ptr(xs), xs.stride(0), l1.cp, st),
                "stem_conv3x3_planes")
        saved = [(None, None, H, W)]
        h, w = H, W
        last = len(net.layers) - 1
        x = None
        for li in range(1, len(net.layers)):
            l = net.layers[li]
            m = B * h * w
            to_planes = not l.pool and li != last
            # a pooled layer nobody differentiates (the frozen conv1_2 / conv2_2) pools and splits in its epilogue: no fp32
            # activation, no pooling pass (ODW_CONV_POOL_EPILOGUE=0: the separate pass, for comparison)
            pooled_out = (l.pool and not l.trainable and l.relu and h % 2 == 0 and w % 2 == 0
                          and os.environ.get("ODW_CONV_POOL_EPILOGUE") != "0")
            if pooled_out:
                y = torch.empty((B * (h // 2) * (w // 2), 2 * l.cout), dtype=torch.bfloat16, device=dev)
                _conv3x3_planes2(lib, xs, m, h, w, l, y, 2, net.zero_page, st)
                saved.append((None, None, h, w))
                xs = y
                h, w = h // 2, w // 2
                continue
            y = torch.empty((m, 2 * l.cout), dtype=torch.bfloat16, device=dev) if to_planes else \
                torch.empty((m, l.cout), dtype=torch.float32, device=dev)
            _conv3x3_planes2(lib, xs, m, h, w, l, y, 1 if to_planes else 0, net.zero_page, st)
            x16 = None
            if l.trainable:
                # the TN / halo weight-gradient kernels read the hi plane in place (row stride 2 cp); the others want it dense
                x16 = xs[:, :l.cp] if _wgrad_reads_strided(l) else xs[:, :l.cp].contiguous()
            pre = None
            if l.pool:
                pre = y if l.trainable else None
                nxt = torch.empty((B * (h // 2) * (w // 2), 2 * l.cout), dtype=torch.bfloat16, device=dev)
                L.check(lib.odw_maxpool2x2_nhwc_f32_planes2(L.ptr(y), B, h, w, l.cout, L.ptr(nxt), nxt.stride(0), st), "maxpool_planes2")
                xs = nxt
            elif to_planes:
                xs = y
            else:
                x = y
            saved.append((x16, pre, h, w))
            if l.pool:
                h, w = h // 2, w // 2
        assert x is not None, "the body's last layer must not be pooled"
        cl = net.layers[-1].cout
        feat = torch.empty((B, cl, h, w), dtype=torch.float32, device=dev)
        L.check(lib.odw_nhwc_f32_to_nchw_f32(L.ptr(x), B, h * w, cl, cl, L.ptr(feat), st), "nhwc_to_nchw_f32")
        ctx.net, ctx.saved_acts, ctx.batch = net, saved, B
        net.last_nhwc = None
        net.last_nhwc_f32 = x
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        return _backward_single_plane(ctx, dfeat)


class VGGBackboneHip(nn.Module):
    """Drop-in for VGG_Base.forward: same parameters, gfx950 kernels."""

    def __init__(self, vgg_base):
        super().__init__()
        self.base = [vgg_base]            # not registered: the parameters stay owned by VGG_Base
        self.layers = _layers_of(vgg_base.features)
        self.zero_page = None
        self.last_nhwc = None
        self._frozen_ready = False
        self._graphs = collections.OrderedDict()        # least recently used first
        self._sightings = {}
        self.graph_stats = {"captures": 0, "replays": 0, "eager": 0, "evictions": 0, "bytes": 0}

    def _prep(self):
        lib = L.lib()
        dev = self.layers[0].conv.weight.device
        if self.zero_page is None:
            self.zero_page = torch.zeros(64, dtype=torch.bfloat16, device=dev)
        first = min(i for i, l in enumerate(self.layers) if l.trainable) if any(l.trainable for l in self.layers) else 99
        mode = P.get_precision()
        mixed = P.split_mode() and not P.bwd_split()      # "bf16x2f": split forward operand, single-plane dgrad operand
        todo = []
        body_p2 = mixed and planes2_body(self)
        for i, l in enumerate(self.layers):
            # frozen layers pack their copies once -- until someone writes their weights in place (a checkpoint loaded
            # after the first forward: copy_ bumps the tensor's version counter)
            if (not l.trainable and self._frozen_ready and l.mode == mode and (l.p2 is None or l.p2 == (body_p2 and i > 0))
                    and getattr(l, "packed_version", None) == l.conv.weight._version):
                continue
            l.packed_version = l.conv.weight._version
            fresh = l.mode != mode
            if P.split_mode() and not mixed:
                # "bf16x3" / "bf16x2": both packed copies as bf16 planes -- rows (co, tap) x [T blocks of Cp] and rows
                # (ci, tap) x [T blocks of Cout] (a handful of small torch passes per layer: the parity modes)
                _, pb = P.patterns("conv")
                wt = l.conv.weight.detach()
                wr = torch.zeros((l.cout, 9, l.cp), dtype=torch.float32, device=dev)
                wr[:, :, :l.cin] = wt.permute(0, 2, 3, 1).reshape(l.cout, 9, l.cin)
                l.wk = P.pack_conv_weight(wr.view(l.cout * 9, l.cp), pb, l.cp, l.cout)
                l.wd = None
                if l.trainable and i > first:
                    wr2 = wt.permute(1, 2, 3, 0).reshape(l.cin * 9, l.cout).contiguous()
                    l.wd = P.pack_conv_weight(wr2, pb, l.cout, l.cin)
                l.mode = mode
                continue
            # "bf16": wk and wd as packed bf16.  "bf16x2f": wk as the forward PLANES (T blocks of Cp per tap), wd as packed
            # bf16 -- every layer's copies in ONE launch (weight_prep_batch_kernel; the per-layer torch passes of the
            # parity modes -- fill, permuted copy, split, pad -- were 36 launches and 0.5 ms of launch gaps at the head of
            # every step)
            p2 = body_p2 and i > 0        # [hi 32 | mid 32] per tap and 32-channel block (T = -2)
            t_blocks = 2 if p2 else (len(P.conv_patterns(l.cp)[1]) if mixed else 1)
            if getattr(l, "p2", None) != p2:
                l.p2, l.wk = p2, None
            if l.wk is None or fresh:
                l.wk = torch.empty((l.cout, _r64(9 * t_blocks * l.cp)), dtype=torch.bfloat16, device=dev)
                l.wd = None
                if l.trainable and i > first:
                    l.wd = torch.empty((l.cin, _r64(9 * l.cout)), dtype=torch.bfloat16, device=dev)
            todo.append(l)
            l.mode = mode
        if todo:
            import ctypes
            n = len(todo)
            key = (mode, tuple(bool(getattr(l, "p2", False)) for l in todo)) + \
                tuple((l.conv.weight.data_ptr(), l.wk.data_ptr(), l.wd.data_ptr() if l.wd is not None else 0) for l in todo)
            if getattr(self, "_prep_key", None) != key:      # the argument arrays change only when a buffer moves
                vp, ia = ctypes.c_void_p * n, ctypes.c_int * n
                pats = [list(P.conv_patterns(l.cp)[1]) if mixed else [] for l in todo]
                args = (vp(*[k[0] for k in key[2:]]), ia(*[l.cout for l in todo]), ia(*[l.cin for l in todo]),
                        ia(*[l.cp for l in todo]), vp(*[k[1] for k in key[2:]]), ia(*[l.wk.stride(0) for l in todo]),
                        vp(*[k[2] or None for k in key[2:]]), ia(*[l.wd.stride(0) if l.wd is not None else 0 for l in todo]),
                        ia(*[-2 if getattr(l, "p2", False) else len(pt) for l, pt in zip(todo, pats)]),
                        (ctypes.c_int * (4 * n))(*[v for pt in pats for v in (pt + [3] * 4)[:4]]))
                self._prep_key, self._prep_args = key, (args, [ctypes.cast(a, ctypes.c_void_p) for a in args])
            L.check(lib.odw_conv_weight_prep_planes_batch(n, *self._prep_args[1], L.stream()), "conv_weight_prep_batch")
        self._frozen_ready = True

    # ---- HIP graphs (engine.build_training_step sets use_graphs): the body's forward (weight packing included) and its
    # backward are two static launch sequences per input shape -- ~50 and ~60 launches that the host issues one ctypes
    # call at a time while the GPU, in the backward half of the step, runs them faster than they arrive.  Captured once
    # per (B, H, W) and replayed: inputs are copied into the graph's static buffers, every tensor the launches touch
    # lives in the graph's memory pool, the gradients land in the parameters' .grad views as in the eager path.
    use_graphs = False
    # data-parallel runs (engine.build_training_step): the backward in this many runs of layers, each announced to
    # on_segment_done(k) when its launches are queued -- its weight gradients start their all-reduce under the rest
    bwd_segments = 1
    on_segment_done = None

    # The cache of captured shapes is BOUNDED: the reference trains multi-scale (configs/voc/*.yaml:33-34: six MIN_SIZE_TRAIN
    # values, free aspect ratios, SIZE_DIVISIBILITY 32, two images per GPU -> hundreds of distinct padded shapes), and every
    # captured pair keeps its whole body's activations in a private pool (~1 GB at VOC sizes).  Policy:
    #   * at most `graph_cache_size` shapes, least recently used evicted (its graphs and pool are released);
    #   * a shape is captured only when it comes back (`graph_min_sightings`): the first time it runs eagerly -- a shape
    #     that never repeats costs nothing beyond its ordinary launches;
    #   * captures are rate-limited by the replays they buy (one capture = a warm-up forward + backward, two captures and a
    #     device synchronisation): never more than 2 + replays / 8, so a workload whose shapes cycle faster than the cache
    #     holds degrades to the eager path instead of re-capturing every step.
    graph_cache_size = int(os.environ.get("ODW_GRAPH_CACHE", "4"))
    graph_min_sightings = 2

    def invalidate_weights(self):
        """The parameters were overwritten from outside the optimiser (a checkpoint): frozen layers re-pack their copies."""
        self._frozen_ready = False

    def _graph_for(self, fn, images):
        """The captured pair for this (function, shape, mode) or None = run this step eagerly."""
        key = (fn.__name__, tuple(images.shape), P.get_precision(), bool(getattr(self, "accumulate", False)),
               int(getattr(self, "bwd_segments", 1)))
        st = self.graph_stats
        g = self._graphs.get(key)
        if g is not None:
            self._graphs.move_to_end(key)
            st["replays"] += 1
            return g
        if len(self._sightings) > 8192:
            self._sightings.clear()
        seen = self._sightings[key] = self._sightings.get(key, 0) + 1
        if self.graph_cache_size < 1 or seen < self.graph_min_sightings or st["captures"] >= 2 + st["replays"] // 8:
            st["eager"] += 1
            return None
        while len(self._graphs) >= self.graph_cache_size:
            torch.cuda.current_stream().synchronize()           # (no replay of the evicted pair may still be in flight)
            old_key, old = self._graphs.popitem(last=False)
            st["evictions"] += 1
            st["bytes"] -= old.bytes
            del old
        before = torch.cuda.memory_allocated(images.device)
        g = self._graphs[key] = _GraphedBody(self, fn, images)
        g.bytes = max(0, torch.cuda.memory_allocated(images.device) - before)
        st["captures"] += 1
        st["bytes"] += g.bytes
        logging.getLogger("od_wscl_amd").info("HIP graphs of the body captured for %s: %.2f GB held (%d shape(s) cached, %.2f GB)",
                                              tuple(images.shape), g.bytes / 1e9, len(self._graphs), st["bytes"] / 1e9)
        return g

    def forward(self, images):
        L.need_gpu(images)
        params = [p for l in self.layers for p in (l.conv.weight, l.conv.bias)]
        fn = (_VGGSplitFn if P.bwd_split() else _VGGMixedFn) if P.split_mode() else _VGGFn
        trainable = any(l.trainable for l in self.layers)
        if (self.use_graphs and torch.is_grad_enabled() and trainable and fn is not _VGGSplitFn
                and all(p.grad is not None for l in self.layers if l.trainable for p in (l.conv.weight, l.conv.bias))
                and getattr(self, "debug", None) is None and os.environ.get("ODW_NO_GRAPHS") != "1"):
            graphed = self._graph_for(fn, images)
        else:
            graphed = None
        if graphed is not None:
            if any(getattr(l, "packed_version", None) != l.conv.weight._version for l in self.layers if not l.trainable):
                with torch.no_grad():       # a frozen layer was rewritten since the capture (which packs trainable layers only):
                    self._prep()            # re-pack into the same buffers the captured launches read
            feat = _GraphedVGGFn.apply(images.float(), self, graphed, *params)
        else:
            with torch.no_grad():
                self._prep()
            feat = fn.apply(images.float(), self, *params)
        feat._odw_nhwc = self.last_nhwc
        feat._odw_nhwc_f32 = getattr(self, "last_nhwc_f32", None) if fn is _VGGMixedFn else None
        # a consumer that produces d(feat) itself (fused ROI pooling) may write it straight into the graph's static input
        feat._odw_grad_out = graphed.dfeat if graphed is not None else None
        return [feat]


class _Ctx(object):
    """Stand-in for the autograd context of the body's Function when its forward / backward run under graph capture."""
    needs_input_grad = ()


class _GraphedBody(object):
    """The forward and the backward of one body for one input shape as two captured HIP graphs."""

    def __init__(self, net, fn, images):
        from ...utils.kernel_timer import kernel_timer as kt
        self.net, self.fn = net, fn
        self.bytes = 0
        dev = images.device
        self.img = torch.empty(tuple(images.shape), dtype=torch.float32, device=dev)
        self.img.copy_(images)
        self.ctx = _Ctx()
        self.ctx.needs_input_grad = (False, False) + (False,) * (2 * len(net.layers))
        was = kt.active
        kt.active = False                              # no event records inside a capture
        seg_cb, net.on_segment_done = getattr(net, "on_segment_done", None), None      # (nor exchanges from the warm-up)
        try:
            # warm-up on a side stream (allocator and lazily-built constants settle), then capture
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                net._prep()
                # (the warm-up doubles as the counting pass: the MFMA work of the two launch sequences, which a replay
                # reports to the kernel timer as ONE region each -- the launches inside a graph carry no events)
                kt.tally, kt.tally_alg = 0.0, 0.0
                feat = fn.forward(self.ctx, self.img, net)
                self.flops_fwd, self.alg_fwd, kt.tally, kt.tally_alg = kt.tally, kt.tally_alg, 0.0, 0.0
                fn.backward(self.ctx, torch.zeros_like(feat))
                self.flops_bwd, self.alg_bwd, kt.tally = kt.tally, kt.tally_alg, None
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.g_fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fwd), torch.no_grad():
                net._prep()
                self.feat = fn.forward(self.ctx, self.img, net)
            self.nhwc = net.last_nhwc
            self.nhwc_f32 = getattr(net, "last_nhwc_f32", None)
            self.dfeat = torch.zeros_like(self.feat)
            # the backward as ONE graph, or (data-parallel runs: net.bwd_segments > 1) one per run of layers, so that the
            # host can start the all-reduce of a run's weight gradients between two replays
            nseg = len(backward_segments(net)) if (fn is not _VGGSplitFn and getattr(net, "bwd_segments", 1) > 1) else 1
            self.g_bwd = []
            for k in range(nseg):
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, pool=self.g_fwd.pool()), torch.no_grad():
                    if nseg == 1:
                        fn.backward(self.ctx, self.dfeat)
                    else:
                        _backward_single_plane(self.ctx, self.dfeat, seg=k)
                self.g_bwd.append(gk)
        finally:
            kt.active = was
            kt.tally = None
            net.on_segment_done = seg_cb


class _GraphedVGGFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, net, graphed, *params):
        graphed.img.copy_(images)
        with kernel_timer.region("VGG body forward (HIP graph)", flops=graphed.flops_fwd, alg=graphed.alg_fwd):
            graphed.g_fwd.replay()
        net.last_nhwc = graphed.nhwc
        net.last_nhwc_f32 = graphed.nhwc_f32
        ctx.graphed = graphed
        # a NEW tensor object over the graph's static buffer every step: a tensor object that is returned twice keeps
        # the hook dictionary of its first autograd node, and Tensor.register_hook then never reaches the new node
        # (the head's early exchange / update, hooked on this tensor, ran in the first step only)
        return graphed.feat.detach()

    @staticmethod
    def backward(ctx, dfeat):
        g = ctx.graphed
        if dfeat.data_ptr() != g.dfeat.data_ptr():      # (the pooling node wrote straight into the buffer: _odw_grad_out)
            g.dfeat.copy_(dfeat)
        with kernel_timer.region("VGG body backward (HIP graph)", flops=g.flops_bwd, alg=g.alg_bwd):
            cb = getattr(g.net, "on_segment_done", None)
            for k, gk in enumerate(g.g_bwd):
                gk.replay()
                if cb is not None and len(g.g_bwd) > 1:
                    cb(k)
        return (None, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)
