"""SupConLossV2 (wetectron/modeling/roi_heads/sim_head/sim_loss.py:44-80) on the fused
gfx950 kernel: forward and backward never materialise the N x N similarity matrix."""
import torch
import torch.nn as nn

from .... import _C


class _SupConV2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, labels, weights, temperature):
        loss, dF = _C.supcon_v2(features, labels, weights, temperature, grad_scale=1.0,
                                need_grad=features.requires_grad)
        ctx.save_for_backward(dF)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dF,) = ctx.saved_tensors
        return dF * g, None, None, None


class SupConLossV2(nn.Module):
    def __init__(self, temperature=0.2):
        super().__init__()
        self.temperature = temperature

    def forward(self, overlaps_enc, score_col, device=None):
        """overlaps_enc: per-class list of (n_c,128) embeddings; score_col: (N,) weights in the
        order they were appended (Q1: NOT re-ordered to match the class-major features)."""
        feats, labels = [], []
        for c, emb in enumerate(overlaps_enc):
            if emb.shape[0] != 0:
                feats.append(emb)
                labels.append(torch.full((emb.shape[0],), c, dtype=torch.int32, device=emb.device))
        features = torch.cat(feats)
        labels = torch.cat(labels)
        return _SupConV2Fn.apply(features, labels, score_col.detach(), self.temperature)
