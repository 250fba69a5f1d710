"""Linear layer of the ROI head (fc6 / fc7 / Sim_Net / predictor GEMMs).

Keeps nn.Linear's parameters (state-dict names unchanged).  `tag` names the layer for the
bench's per-kernel timing."""
import torch
import torch.nn.functional as F
from torch import nn

from ..utils.kernel_timer import kernel_timer


class Linear(nn.Linear):
    tag = None

    def forward(self, x):
        if self.tag is None or not kernel_timer.enabled:
            return F.linear(x, self.weight, self.bias)
        m = x.shape[0]
        flops = 2.0 * m * self.in_features * self.out_features
        with kernel_timer.region(self.tag + "_fwd", flops=flops):
            return F.linear(x, self.weight, self.bias)
