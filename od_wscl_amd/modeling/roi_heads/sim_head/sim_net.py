"""Sim_Net (wetectron/modeling/roi_heads/sim_head/sim_net.py:7-26): 4096 -> 4096 -> 128,
L2-normalised rows.  On the gfx950 back end both Linears run on the MFMA GEMM (ReLU fused into
the first epilogue, fp32 output from the second so the normalisation sees full precision)."""
import torch
import torch.nn as nn

from .... import _lib as L
from ....layers.linear import Linear


class _L2NormRows(torch.autograd.Function):
    """F.normalize(x, dim=1) of the (R, 128) embeddings as one launch each way (csrc/head_aux.hip)."""

    @staticmethod
    def forward(ctx, x, eps):
        x = x.contiguous()
        R, D = x.shape
        y = torch.empty_like(x)
        norm = torch.empty(R, dtype=torch.float32, device=x.device)
        L.check(L.lib().odw_l2norm_rows(L.ptr(x), R, D, eps, L.ptr(y), L.ptr(norm), L.stream()), "l2norm_rows")
        ctx.save_for_backward(y, norm)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, g):
        y, norm = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty_like(g)
        L.check(L.lib().odw_l2norm_rows_bwd(L.ptr(g), L.ptr(y), L.ptr(norm), y.shape[0], y.shape[1], ctx.eps, L.ptr(dx),
                                            L.stream()), "l2norm_rows_bwd")
        return dx, None


class Sim_Net(nn.Module):
    def __init__(self, config, in_dim):
        super().__init__()
        self.mlp = nn.Sequential(Linear(in_dim, in_dim), nn.ReLU(inplace=True), Linear(in_dim, 128))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)

    kept = None        # (mlp[0] output, mlp[2] output) of the last no-autograd evaluation with keep=True

    def forward(self, roi_feat, keep=False):
        rows = getattr(roi_feat, "_odw_reuse_rows", None)
        if rows is not None and self.kept is not None:
            # rows of the evaluation that already ran over the whole clean pass: gathered and re-attached to the graph
            # (layers.Linear.reuse) instead of two more GEMMs; the normalisation of a few hundred rows is redone
            h = self.mlp[0].reuse(roi_feat, self.kept[0], rows, relu=True)
            e = self.mlp[2].reuse(h, self.kept[1], rows)
            return _L2NormRows.apply(e.float(), 1e-12)
        h = self.mlp[0].fused(roi_feat, relu=True)
        e = self.mlp[2].fused(h, out_f32=True)
        if keep:
            self.kept = (h.detach(), e.detach())
        return _L2NormRows.apply(e, 1e-12)
