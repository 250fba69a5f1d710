"""Device-side counter-based randomness with autograd support (kernels: csrc/rng.hip).

One logical draw = one stream id, handed out in call order, exactly like the
checker's `oracle.hotpath_ref.Rand` and the generator injected into the
reference when the golden vectors were made -- so a parity run sees the same
dropout masks / DropBlock centres / noise as the reference did."""
import torch

from .. import _lib as L
from . import rng as _rng


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k0, k1, p):
        ctx.key = (k0, k1, p)
        x = x.contiguous()
        out = torch.empty_like(x)
        L.check(L.lib().odw_dropout(L.ptr(x), L.ptr(out), x.numel(), k0, k1, p, L.stream()), "dropout")
        return out

    @staticmethod
    def backward(ctx, g):
        k0, k1, p = ctx.key
        g = g.contiguous()
        out = torch.empty_like(g)   # same mask, re-derived from the counter
        L.check(L.lib().odw_dropout(L.ptr(g), L.ptr(out), g.numel(), k0, k1, p, L.stream()), "dropout")
        return out, None, None, None


class _NoiseMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k0, k1):
        ctx.key = (k0, k1)
        x = x.contiguous()
        out = torch.empty_like(x)
        L.check(L.lib().odw_noise_mul(L.ptr(x), L.ptr(out), x.numel(), k0, k1, L.stream()), "noise_mul")
        return out

    @staticmethod
    def backward(ctx, g):
        k0, k1 = ctx.key
        g = g.contiguous()
        out = torch.empty_like(g)   # d(x + z x)/dx = 1 + z: the same kernel applied to the gradient
        L.check(L.lib().odw_noise_mul(L.ptr(g), L.ptr(out), g.numel(), k0, k1, L.stream()), "noise_mul")
        return out, None, None


def dropout_with_segments(x, p, segs):
    """Dropout of a row-stacked activation: rows [segs[i][0], segs[i+1][0]) use key segs[i][1:]
    with element index relative to the segment's first row."""
    if len(segs) == 1:
        return _Dropout.apply(x.float(), segs[0][1], segs[0][2], float(p))
    parts = []
    bounds = [s[0] for s in segs] + [x.shape[0]]
    for i, s in enumerate(segs):
        parts.append(_Dropout.apply(x[bounds[i]:bounds[i + 1]].float(), s[1], s[2], float(p)))
    return torch.cat(parts, dim=0)


class DeviceRand(object):
    def __init__(self, seed, first_stream=1 << 20, device="cuda"):
        self.s = _rng.Streams(seed, first_stream)
        self.device = device

    def _key(self):
        return _rng.stream_key(self.s.seed, self.s.take())

    def key(self):
        """(k0, k1) of the next logical draw (for epilogue-fused dropout)."""
        return self._key()

    def uniform(self, shape):
        k0, k1 = self._key()
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        L.check(L.lib().odw_rng_uniform(L.ptr(out), out.numel(), k0, k1, 0, L.stream()), "rng_uniform")
        return out

    def normal(self, shape):
        k0, k1 = self._key()
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        L.check(L.lib().odw_rng_normal(L.ptr(out), out.numel(), k0, k1, L.stream()), "rng_normal")
        return out

    def dropout(self, x, p=0.5):
        k0, k1 = self._key()
        L.need_gpu(x)
        return _Dropout.apply(x.float(), k0, k1, float(p))

    def noise_mul(self, x):
        k0, k1 = self._key()
        L.need_gpu(x)
        return _NoiseMul.apply(x.float(), k0, k1)
