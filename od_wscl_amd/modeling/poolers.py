"""Pooler (wetectron/modeling/poolers.py:46-128), single feature level (the only case the
OD-WSCL configs use): list[BoxList] -> (R,5) rois -> ROIPool / ROIAlign."""
import torch
from torch import nn

from ..layers import ROIAlign, ROIPool


class Pooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio, method="ROIPool"):
        super().__init__()
        if len(scales) != 1:
            raise NotImplementedError("multi-level (FPN) pooling is outside the OD-WSCL hot path")
        if method == "ROIPool":
            self.poolers = nn.ModuleList([ROIPool(output_size, spatial_scale=scales[0])])
        elif method == "ROIAlign":
            self.poolers = nn.ModuleList([ROIAlign(output_size, spatial_scale=scales[0], sampling_ratio=sampling_ratio)])
        else:
            raise ValueError("please use valid pooler function")
        self.output_size = output_size

    @staticmethod
    def convert_to_roi_format(boxes):
        """rows [batch_index, x1, y1, x2, y2] (poolers.py:85-96)."""
        parts = []
        for i, b in enumerate(boxes):
            idx = torch.full((len(b), 1), float(i), dtype=b.bbox.dtype, device=b.bbox.device)
            parts.append(torch.cat([idx, b.bbox], dim=1))
        return torch.cat(parts, dim=0)

    def forward(self, x, boxes):
        return self.poolers[0](x[0], self.convert_to_roi_format(boxes))
