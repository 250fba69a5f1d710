"""GeneralizedRCNN with precomputed proposals (wetectron/modeling/detector/generalized_rcnn.py:23-97):
backbone -> ROI weak head -> (loss dict, accuracy dict)."""
import torch
from torch import nn

from ..backbone import build_backbone
from ..roi_heads.weak_head.weak_head import build_roi_weak_head


class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg.MODEL.FASTER_RCNN:
            raise NotImplementedError("RPN models are outside the OD-WSCL hot path (FASTER_RCNN: False)")
        self.backbone = build_backbone(cfg)
        self.roi_heads = build_roi_weak_head(cfg, self.backbone.out_channels)

    def hip_body(self):
        """The HIP rendition of self.backbone.body (created on first use: it allocates device constants)."""
        hip = getattr(self, "backbone_hip", None)
        if hip is None:
            body = self.backbone.body
            if type(body).__name__ == "VGG_Base":
                from ..backbone.vgg16_hip import VGGBackboneHip
                hip = VGGBackboneHip(body)
            else:
                from ..backbone.resnet_hip import ResNetBackboneHip
                hip = ResNetBackboneHip(body)
            self.backbone_hip = hip
        return hip

    def forward(self, images, targets=None, rois=None, model_cdb=None, iteration=None, rand=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        if rois is None or rois[0] is None:
            raise ValueError("precomputed proposals (rois) are required")
        if rand is not None:
            self.roi_heads.set_rand(rand)
        # the body runs on the gfx950 kernels (NHWC implicit-GEMM convolutions, modeling/backbone/vgg16_hip.py /
        # resnet_hip.py); `self.backbone` only owns the reference-named parameters -- there is no library path
        features = self.hip_body()(images.tensors)
        x, result, losses, accuracy = self.roi_heads(features, rois, targets, model_cdb, iteration)
        if self.training:
            return (losses if isinstance(losses, dict) else dict(losses)), accuracy
        return result
