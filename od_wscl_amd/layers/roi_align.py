"""ROIAlign operator -- same API as wetectron/layers/roi_align.py:11-68
(legacy un-aligned sampling; sampling_ratio <= 0 = adaptive grid)."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _C


class _ROIAlign(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        ph, pw = _pair(output_size)
        ctx.geom = (ph, pw, float(spatial_scale), int(sampling_ratio), tuple(input.shape))
        ctx.save_for_backward(roi)
        return _C.roi_align_forward(input, roi, spatial_scale, ph, pw, sampling_ratio)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (roi,) = ctx.saved_tensors
        ph, pw, scale, sr, (bs, ch, h, w) = ctx.geom
        grad_input = _C.roi_align_backward(grad_output, roi, scale, ph, pw, bs, ch, h, w, sr)
        return grad_input, None, None, None, None


roi_align = _ROIAlign.apply


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        return roi_align(input.float(), rois.float(), self.output_size, self.spatial_scale,
                         self.sampling_ratio)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)
