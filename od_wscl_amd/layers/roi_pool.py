"""ROIPool operator -- same API as wetectron/layers/roi_pool.py:11-64
(`ROIPool(output_size, spatial_scale)`, `roi_pool(input, rois, output_size,
spatial_scale)`), backed by the plane-resident gfx950 kernels."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _C


class _ROIPool(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale):
        ph, pw = _pair(output_size)
        ctx.geom = (ph, pw, float(spatial_scale), tuple(input.shape))
        output, argmax = _C.roi_pool_forward(input, roi, spatial_scale, ph, pw)
        # the reference also saves `input` (roi_pool.py:20) only to read its sizes
        ctx.save_for_backward(roi, argmax)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        roi, argmax = ctx.saved_tensors
        ph, pw, scale, (bs, ch, h, w) = ctx.geom
        grad_input = _C.roi_pool_backward(grad_output, None, roi, argmax, scale, ph, pw, bs, ch, h, w)
        return grad_input, None, None, None


roi_pool = _ROIPool.apply


class ROIPool(nn.Module):
    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale

    def forward(self, input, rois):
        # amp.float_function in the reference: the op always sees fp32
        return roi_pool(input.float(), rois.float(), self.output_size, self.spatial_scale)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale)
