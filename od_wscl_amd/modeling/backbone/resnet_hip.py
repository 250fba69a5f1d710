"""ResNet-C5 bodies (R-50-C5 / R-101-C5, wetectron/modeling/backbone/resnet.py:84-148,258-406) on the gfx950 kernels.

NHWC bf16 activations (n_pix x C matrices) end to end:
  * every 1x1 convolution IS a Linear over the pixel rows -> the MFMA GEMM of the head (csrc/gemm_bf16.hip through
    gemm.fused_linear: forward, input gradient, weight gradient) with bias + ReLU in the epilogue;
  * the 3x3 convolutions run on the implicit-GEMM kernel of the VGG body (conv3x3_glds_kernel; dgrad = the mirrored
    kernel, wgrad = transposed im2col + split-K GEMM);
  * the frozen batch-norms (layers/batch_norm.py:6-31) are folded: y = conv(x, W * s[co]) + t[co] -- scale into the
    bf16 weight copy, shift as the epilogue bias; the weight gradient flows back through the scale by autograd;
  * residual junction relu(y3 + identity) and its gradient: csrc/resnet_aux.hip; the stride of the first 1x1 of
    layer2.0 / layer3.0 (STRIDE_IN_1X1) is a row subsampling of the NHWC matrix shared by conv1 and the projection;
  * stem (7x7/2 + BN + ReLU, 3x3/2 max pool): csrc/resnet_aux.hip, forward only -- the stem and layer1 are frozen in
    every shipped config (FREEZE_CONV_BODY_AT 2) and nothing below layer2 receives a gradient.
The parameters stay the nn.Conv2d weights / FrozenBatchNorm2d buffers of the reference-named modules."""
import os

import torch
from torch import nn

from ... import _lib as L
from ... import gemm
from ... import precision as P
from ...utils.kernel_timer import kernel_timer
from .vgg16_hip import _conv3x3, _r64, conv_wgrad


def _fold(bn):
    """scale, shift of a FrozenBatchNorm2d (batch_norm.py:23-31)."""
    scale = bn.weight * bn.running_var.rsqrt()
    return scale.float().contiguous(), (bn.bias - bn.running_mean * scale).float().contiguous()


class _Conv3x3Fn(torch.autograd.Function):
    """y = relu(conv3x3(x, w) + shift) on (B*H*W, C) bf16 rows; w = the folded fp32 (Co, Ci, 3, 3) weight."""

    @staticmethod
    def forward(ctx, x, w, shift, geom, zero_page, need_dx):
        lib, st = L.lib(), L.stream()
        B, H, W = geom
        co, ci = w.shape[0], w.shape[1]
        m = B * H * W
        wk = torch.empty((co, _r64(9 * ci)), dtype=torch.bfloat16, device=x.device)
        wd = torch.empty((ci, _r64(9 * co)), dtype=torch.bfloat16, device=x.device) if need_dx else None
        L.check(lib.odw_conv_weight_prep(L.ptr(w.detach().contiguous()), co, ci, ci, L.ptr(wk), wk.stride(0), L.ptr(wd),
                                         wd.stride(0) if wd is not None else 0, st), "conv_weight_prep")
        y = torch.empty((m, co), dtype=torch.bfloat16, device=x.device)
        _conv3x3(lib, x, m, H, W, ci, 1, 0, wk, co, y, shift, True, None, 0, zero_page, st, 2.0 * m * co * 9 * ci)
        ctx.save_for_backward(x, y, wd, zero_page)
        ctx.dims = (B, H, W, co, ci)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, wd, zero_page = ctx.saved_tensors
        return _conv3x3_backward_single_plane(ctx, dy, x, y, wd, zero_page, torch.bfloat16)


def _conv3x3_backward_single_plane(ctx, dy, x, y, wd, zero_page, dx_dtype):
    """Backward of a 3x3 layer with one bf16 plane per operand: x = the layer input as NHWC bf16 rows, y = the saved
    output (bf16, or fp32 after a split-precision forward) the ReLU mask is re-derived from."""
    B, H, W, co, ci = ctx.dims
    lib, st = L.lib(), L.stream()
    m = B * H * W
    m64 = _r64(m)
    dy = dy.contiguous()
    dz = torch.empty((m, _r64(co)), dtype=torch.bfloat16, device=dy.device)
    dzt = torch.empty((co, m64), dtype=torch.bfloat16, device=dy.device)
    flags = (1 if dy.dtype == torch.float32 else 0) | (2 if y.dtype == torch.float32 else 0)
    L.check(lib.odw_linear_bwd_prep(L.ptr(dy), flags, dy.stride(0), L.ptr(y), y.stride(0),
                                    m, co, 1.0, L.ptr(dz), dz.stride(0), L.ptr(dzt), m64, None, st), "conv bwd prep")
    dw = None
    if ctx.needs_input_grad[1] and ci >= 128 and ci & (ci - 1) == 0 and co % 8 == 0 and os.environ.get("ODW_CONV_WGRAD_TN") != "0":
        # dZ and the layer input as they are: K-major operands, transposed fragment reads (odw_conv_wgrad_tn)
        dw = torch.empty((co, ci, 3, 3), dtype=torch.float32, device=dy.device)
        ws_bytes = lib.odw_conv_wgrad_tn_workspace(co, ci, m)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device)
        with kernel_timer.region("conv wgrad TN split-K+reduce", flops=2.0 * m * co * 9 * ci):
            L.check(lib.odw_conv_wgrad_tn(L.ptr(dz), dz.stride(0), L.ptr(x), m, H, W, ci, 1, co, ci, L.ptr(dw), 0,
                                          L.ptr(zero_page), L.ptr(ws), ws_bytes, st), "conv_wgrad_tn")
    elif ctx.needs_input_grad[1]:
        colt = torch.empty((9 * ci, m64), dtype=torch.bfloat16, device=dy.device)
        L.check(lib.odw_im2col_t_bf16(L.ptr(x), m, H, W, ci, 1, L.ptr(colt), m64, st), "im2col_t")
        dw = torch.empty((co, ci, 3, 3), dtype=torch.float32, device=dy.device)
        conv_wgrad(lib, dzt, colt, co, ci, ci, m, dw, st)
    dx = None
    if ctx.needs_input_grad[0] and wd is not None:
        dx = torch.empty((m, ci), dtype=dx_dtype, device=dy.device)
        dzc = dz if dz.shape[1] == co else dz[:, :co].contiguous()
        _conv3x3(lib, dzc, m, H, W, co, 1, 1, wd, ci, dx, None, False, None, 0, zero_page, st, 2.0 * m * co * 9 * ci)
    return dx, dw, None, None, None, None


class _SplitConv3x3Fn(torch.autograd.Function):
    """The same layer in a split precision mode: fp32 rows in and out; the input (and, backward, dZ) are laid out as
    bf16 planes along the channel axis, the weights along their (tap, channel) axis (csrc/split.hip), and the unchanged
    implicit-GEMM kernel runs over T*C channels with an fp32 epilogue (see vgg16_hip._VGGSplitFn)."""

    @staticmethod
    def forward(ctx, x, w, shift, geom, zero_page, need_dx):
        lib, st = L.lib(), L.stream()
        B, H, W = geom
        co, ci = w.shape[0], w.shape[1]
        m = B * H * W
        pa, pb = P.patterns("conv")
        T = len(pa)
        wd32 = w.detach()
        wk = P.pack_conv_weight(wd32.permute(0, 2, 3, 1).reshape(co * 9, ci).contiguous(), pb, ci, co)
        wd = P.pack_conv_weight(wd32.permute(1, 2, 3, 0).reshape(ci * 9, co).contiguous(), pb, co, ci) if need_dx else None
        x = x.contiguous()
        xs = P.split_rows(x, pa, ci)
        y = torch.empty((m, co), dtype=torch.float32, device=x.device)
        _conv3x3(lib, xs, m, H, W, T * ci, 1, 0, wk, co, y, shift, True, None, 0, zero_page, st, 2.0 * m * co * 9 * ci * T)
        ctx.save_for_backward(x, y, wd, zero_page)
        ctx.dims = (B, H, W, co, ci)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, wd, zero_page = ctx.saved_tensors
        B, H, W, co, ci = ctx.dims
        lib, st = L.lib(), L.stream()
        m = B * H * W
        m64 = _r64(m)
        pa, _ = P.patterns("conv")
        ga, gb = P.patterns("gemm")
        T, Tg = len(pa), len(ga)
        dy = dy.contiguous().float()
        dz = P.bwd_mask(dy, y, 1.0)
        dw = None
        if ctx.needs_input_grad[1]:
            dzt = P.split_cols(dz, ga, m64)
            planes = [P.split_rows(x, (p,), ci) for p in (0, 1, 2)]
            colt = torch.empty((9 * ci, Tg * m64), dtype=torch.bfloat16, device=dy.device)
            for t, pl in enumerate(gb):
                L.check(lib.odw_im2col_t_bf16_part(L.ptr(planes[pl]), m, H, W, ci, 1, L.ptr(colt[:, t * m64:]), Tg * m64, m64, st),
                        "im2col_t")
            dw = torch.empty((co, ci, 3, 3), dtype=torch.float32, device=dy.device)
            conv_wgrad(lib, dzt, colt, co, ci, ci, Tg * m64, dw, st)
        dx = None
        if ctx.needs_input_grad[0] and wd is not None:
            dx = torch.empty((m, ci), dtype=torch.float32, device=dy.device)
            dzs = P.split_rows(dz, pa, co)
            _conv3x3(lib, dzs, m, H, W, T * co, 1, 1, wd, ci, dx, None, False, None, 0, zero_page, st, 2.0 * m * co * 9 * ci * T)
        return dx, dw, None, None, None, None


class _MixedConv3x3Fn(torch.autograd.Function):
    """Precision mode "bf16x2f": the forward of _SplitConv3x3Fn (fp32 rows in and out, plane products), the backward of
    _Conv3x3Fn (one bf16 plane per operand; the input gradient comes back as fp32 rows)."""

    @staticmethod
    def forward(ctx, x, w, shift, geom, zero_page, need_dx):
        lib, st = L.lib(), L.stream()
        B, H, W = geom
        co, ci = w.shape[0], w.shape[1]
        m = B * H * W
        pa, pb = P.conv_patterns(ci)
        T = len(pa)
        wd32 = w.detach().contiguous()
        wk = P.pack_conv_weight(wd32.permute(0, 2, 3, 1).reshape(co * 9, ci).contiguous(), pb, ci, co)
        wd = None
        if need_dx:
            wd = torch.empty((ci, _r64(9 * co)), dtype=torch.bfloat16, device=x.device)
            L.check(lib.odw_conv_weight_prep(L.ptr(wd32), co, ci, ci, None, 0, L.ptr(wd), wd.stride(0), st), "conv_weight_prep")
        x = x.contiguous()
        xs = P.split_rows(x, pa, ci)
        y = torch.empty((m, co), dtype=torch.float32, device=x.device)
        _conv3x3(lib, xs, m, H, W, T * ci, 1, 0, wk, co, y, shift, True, None, 0, zero_page, st, 2.0 * m * co * 9 * ci * T)
        x16 = P.split_rows(x, (0,), ci) if ctx.needs_input_grad[1] else None
        ctx.save_for_backward(x16, y, wd, zero_page)
        ctx.dims = (B, H, W, co, ci)
        return y

    @staticmethod
    def backward(ctx, dy):
        x16, y, wd, zero_page = ctx.saved_tensors
        return _conv3x3_backward_single_plane(ctx, dy, x16, y, wd, zero_page, torch.float32)


class _AddReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        fn = L.lib().odw_add_relu_f32 if a.dtype == torch.float32 else L.lib().odw_add_relu_bf16
        L.check(fn(L.ptr(a), L.ptr(b), L.ptr(out), a.numel(), L.stream()), "add_relu")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, = ctx.saved_tensors
        dout = dout.contiguous().to(out.dtype)
        g = torch.empty_like(out)
        fn = L.lib().odw_relu_bwd_f32 if out.dtype == torch.float32 else L.lib().odw_relu_bwd_bf16
        L.check(fn(L.ptr(dout), L.ptr(out), L.ptr(g), out.numel(), L.stream()), "relu_bwd")
        return g, g


class _ToNCHW(torch.autograd.Function):
    """(B*H*W, C) bf16 rows -> (B, C, H, W) fp32, the layout the pooling operators take; backward the other way."""

    @staticmethod
    def forward(ctx, x, geom):
        B, H, W = geom
        C = x.shape[1]
        feat = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        if x.dtype == torch.float32:
            L.check(L.lib().odw_nhwc_f32_to_nchw_f32(L.ptr(x.contiguous()), B, H * W, C, C, L.ptr(feat), L.stream()), "nhwc_to_nchw")
        else:
            L.check(L.lib().odw_nhwc_bf16_to_nchw_f32(L.ptr(x.contiguous()), B, H * W, C, L.ptr(feat), L.stream()), "nhwc_to_nchw")
        ctx.geom = geom
        ctx.f32 = x.dtype == torch.float32
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        B, H, W = ctx.geom
        C = dfeat.shape[1]
        if ctx.f32:
            dx = torch.empty((B * H * W, C), dtype=torch.float32, device=dfeat.device)
            L.check(L.lib().odw_nchw_f32_to_nhwc_f32(L.ptr(dfeat.contiguous().float()), B, H * W, C, C, L.ptr(dx), L.stream()),
                    "nchw_to_nhwc")
            return dx, None
        dx = torch.empty((B * H * W, C), dtype=torch.bfloat16, device=dfeat.device)
        L.check(L.lib().odw_nchw_f32_to_nhwc_bf16(L.ptr(dfeat.contiguous().float()), B, H * W, C, C, L.ptr(dx), L.stream()),
                "nchw_to_nhwc")
        return dx, None


class _Folded(object):
    """Per convolution: the folded batch-norm constants, and for frozen convolutions the bf16 operand copies."""
    __slots__ = ("scale", "shift", "shadow", "key")


class ResNetBackboneHip(nn.Module):
    """Drop-in for the ResNet body's forward (model.backbone.body): same parameters and buffers, gfx950 kernels."""

    def __init__(self, body):
        super().__init__()
        self.base = [body]              # not registered: parameters / buffers stay owned by the reference-named modules
        self.cache = {}
        self.zero_page = None

    # ---- constants ------------------------------------------------------------------------------------------
    def _const(self, conv, bn):
        key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
               conv.weight._version if not conv.weight.requires_grad else -1)
        f = self.cache.get(id(conv))
        if f is None or f.key != key:
            f = _Folded()
            with torch.no_grad():
                f.scale, f.shift = _fold(bn)
            f.shadow, f.key = None, key
            self.cache[id(conv)] = f
        return f

    def _w1x1(self, conv, f):
        """(Co, Ci) folded weight of a 1x1 convolution and its bf16 operand copies (cached while frozen)."""
        co, ci = conv.weight.shape[:2]
        if not conv.weight.requires_grad:
            if f.shadow is None:
                with torch.no_grad():
                    w = (conv.weight.reshape(co, ci) * f.scale[:, None]).contiguous()
                f.shadow = (w, gemm.Shadow(w))
            return f.shadow
        w = conv.weight.reshape(co, ci) * f.scale[:, None]
        return w, gemm.Shadow(w)

    # ---- layers ---------------------------------------------------------------------------------------------
    def _conv1x1(self, x, conv, bn, relu):
        f = self._const(conv, bn)
        w, sh = self._w1x1(conv, f)
        return gemm.fused_linear(x, w, f.shift, sh, relu=relu)

    def _conv3x3(self, x, conv, bn, geom, need_dx):
        f = self._const(conv, bn)
        assert conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
        w = conv.weight * f.scale[:, None, None, None]
        fn = (_SplitConv3x3Fn if P.bwd_split() else _MixedConv3x3Fn) if P.split_mode() else _Conv3x3Fn
        return fn.apply(x, w, f.shift, geom, self.zero_page, need_dx)

    @staticmethod
    def _subsample(x, geom, stride):
        if stride == 1:
            return x, geom
        B, H, W = geom
        xs = x.reshape(B, H, W, x.shape[1])[:, ::stride, ::stride, :]
        Hs, Ws = xs.shape[1], xs.shape[2]
        return xs.reshape(B * Hs * Ws, x.shape[1]), (B, Hs, Ws)

    def _block(self, x, geom, blk, need_dx):
        stride = blk.conv1.stride[0]
        xs, g2 = self._subsample(x, geom, stride)
        a1 = self._conv1x1(xs, blk.conv1, blk.bn1, True)
        a2 = self._conv3x3(a1, blk.conv2, blk.bn2, g2, True)
        y3 = self._conv1x1(a2, blk.conv3, blk.bn3, False)
        if blk.downsample is not None:
            assert blk.downsample[0].stride[0] == stride
            idn = self._conv1x1(xs, blk.downsample[0], blk.downsample[1], False)
        else:
            idn = x
        return _AddReLU.apply(y3, idn), g2

    def forward(self, images):
        L.need_gpu(images)
        body = self.base[0]
        lib, st = L.lib(), L.stream()
        dev = images.device
        if self.zero_page is None:
            self.zero_page = torch.zeros(64, dtype=torch.bfloat16, device=dev)
        images = images.float().contiguous()
        B, _, H, W = images.shape
        with torch.no_grad():
            stem = body.stem
            f = self._const(stem.conv1, stem.bn1)
            co = stem.conv1.weight.shape[0]
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            act = P.act_dtype()
            f32 = act == torch.float32
            s = torch.empty((B * Ho * Wo, co), dtype=act, device=dev)
            stem_fn = lib.odw_stem_conv7x7_bn_relu_f32 if f32 else lib.odw_stem_conv7x7_bn_relu
            L.check(stem_fn(L.ptr(images), L.ptr(stem.conv1.weight.detach().float().contiguous()),
                            L.ptr(f.scale), L.ptr(f.shift), B, H, W, co, L.ptr(s), st), "stem")
            Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
            x = torch.empty((B * Hp * Wp, co), dtype=act, device=dev)
            pool_fn = lib.odw_maxpool3x3s2_nhwc_f32 if f32 else lib.odw_maxpool3x3s2_nhwc_bf16
            L.check(pool_fn(L.ptr(s), B, Ho, Wo, co, L.ptr(x), st), "maxpool3x3s2")
        geom = (B, Hp, Wp)
        seen_trainable = False
        for name in body.stages:
            for blk in getattr(body, name):
                trainable = any(p.requires_grad for p in blk.parameters())
                if trainable and torch.is_grad_enabled():
                    x, geom = self._block(x, geom, blk, need_dx=seen_trainable)
                    seen_trainable = True
                else:
                    with torch.no_grad():
                        x, geom = self._block(x, geom, blk, need_dx=False)
        feat = _ToNCHW.apply(x, geom)
        feat._odw_nhwc = x.detach() if x.dtype == torch.bfloat16 else None     # the NHWC bf16 map: the fused ROI pooling reads it
        feat._odw_nhwc_f32 = x.detach().contiguous() if x.dtype == torch.float32 else None   # ("bf16x2f": the fp32 one)
        return [feat]
