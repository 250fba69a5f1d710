"""MISTPredictor (wetectron/modeling/roi_heads/weak_head/roi_weak_predictors.py:112-187):
cls / det / three refinement classifiers / three box regressors on the 4096-d ROI feature.

The eight Linear layers keep the reference's parameter names, but are evaluated as ONE
GEMM against the row-concatenated weight (N = 5C + 3*4C = 357 for VOC): one pass over
the (P,4096) activations instead of eight."""
import torch
import torch.nn.functional as F
from torch import nn

from ... import registry
from .... import gemm

_HEADS = ("cls_score", "det_score", "ref1", "bbox_pred1", "ref2", "bbox_pred2", "ref3", "bbox_pred3")


@registry.ROI_WEAK_PREDICTOR.register("MISTPredictor")
class MISTPredictor(nn.Module):
    def __init__(self, config, in_channels):
        super().__init__()
        c = config.MODEL.ROI_BOX_HEAD.NUM_CLASSES
        nbox = (2 if config.MODEL.CLS_AGNOSTIC_BBOX_REG else c) * 4
        for name in _HEADS:
            setattr(self, name, nn.Linear(in_channels, nbox if name.startswith("bbox") else c))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0, std=0.001)
                nn.init.constant_(m.bias, 0)

    head_names = _HEADS
    _fused = None

    def set_fused(self, w_cat, b_cat, shadow):
        """engine.FlatSGD lays the 8 heads out back to back in its flat buffers and hands back the ONE
        (5C+12C) x K weight / bias they form there (leaf tensors aliasing the heads' storage and gradients)."""
        self._fused = (w_cat, b_cat, shadow)

    def forward(self, x, proposals):
        assert x.dim() == 2
        heads = [getattr(self, n) for n in _HEADS]
        if not x.is_cuda:
            raise RuntimeError("MISTPredictor: tensor is not on the GPU -- the hot path has no CPU implementation")
        if self._fused is not None and self.training:
            w_cat, b_cat, shadow = self._fused
            out = gemm.fused_linear(x, w_cat, b_cat, shadow, out_f32=True, tag="predictor")
        else:
            w = torch.cat([h.weight for h in heads], dim=0)
            b = torch.cat([h.bias for h in heads], dim=0)
            out = gemm.fused_linear(x, w, None, gemm.Shadow(w), out_f32=True, tag="predictor") + b
        out = out.split([h.out_features for h in heads], dim=1)
        cls, det, r1, b1, r2, b2, r3, b3 = out
        if not self.training:       # roi_weak_predictors.py:167-181
            cls = F.softmax(cls, dim=1)
            det = torch.cat([F.softmax(d, dim=0) for d in det.split([len(p) for p in proposals])], dim=0)
            r1, r2, r3 = F.softmax(r1, dim=1), F.softmax(r2, dim=1), F.softmax(r3, dim=1)
        return cls, det, [r1, r2, r3], [b1, b2, b3]


def make_roi_weak_predictor(cfg, in_channels):
    return registry.ROI_WEAK_PREDICTOR[cfg.MODEL.ROI_WEAK_HEAD.PREDICTOR](cfg, in_channels)
