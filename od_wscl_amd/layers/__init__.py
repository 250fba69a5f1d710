"""Operator API mirroring wetectron/layers/__init__.py:4-46 for the hot path."""
from .misc import Conv2d, FrozenBatchNorm2d
from .nms import nms, nms_torchvision
from .roi_align import ROIAlign, roi_align
from .roi_pool import ROIPool, roi_pool
from .smooth_l1_loss import smooth_l1_loss

__all__ = ["nms", "nms_torchvision", "roi_align", "ROIAlign", "roi_pool", "ROIPool", "smooth_l1_loss",
           "Conv2d", "FrozenBatchNorm2d"]
