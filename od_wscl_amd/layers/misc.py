"""Conv2d / FrozenBatchNorm2d with the reference's names
(wetectron/layers/misc.py:31-48, layers/batch_norm.py:6-31)."""
import torch
from torch import nn


class Conv2d(torch.nn.Conv2d):
    """torch.nn.Conv2d; the reference's subclass only adds an empty-batch path
    (layers/misc.py:31-48) which modern torch handles natively."""


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d whose statistics and affine parameters are constants
    (layers/batch_norm.py:6-31): y = x * (w * rsqrt(var)) + (b - mean * w * rsqrt(var))."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def forward(self, x):
        scale = self.weight * self.running_var.rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)
