"""Conv2d / FrozenBatchNorm2d with the reference's names
(wetectron/layers/misc.py:31-48, layers/batch_norm.py:6-31)."""
import contextlib

import torch
from torch import nn

_LIBRARY_REFERENCE = [0]


@contextlib.contextmanager
def library_reference():
    """Tests and tools/ only: inside this context Conv2d.forward runs torch's own convolution (MIOpen on the GPU) -- the
    plain PyTorch REFERENCE a HIP kernel is compared with.  The product never enters it."""
    _LIBRARY_REFERENCE[0] += 1
    try:
        yield
    finally:
        _LIBRARY_REFERENCE[0] -= 1


class Conv2d(torch.nn.Conv2d):
    """The reference's Conv2d (layers/misc.py:31-48: torch.nn.Conv2d plus an empty-batch path) as the OWNER of a
    convolution's parameters under the reference's state-dict names.  In this package every convolution of a body runs
    inside GeneralizedRCNN.hip_body() -- the implicit-GEMM 3x3 / 1x1-as-GEMM / 7x7-stem kernels of csrc/ reading these
    parameters -- so a direct call has no HIP rendition and there is deliberately NO library (MIOpen) path to fall back
    to: it raises.  (A comparison against torch's convolution opts in with `library_reference()`.)"""

    def forward(self, x):
        if not _LIBRARY_REFERENCE[0]:
            raise RuntimeError("od_wscl_amd.layers.Conv2d has no standalone forward: the bodies' convolutions run on the gfx950 "
                               "kernels through GeneralizedRCNN.hip_body() (modeling/backbone/vgg16_hip.py, resnet_hip.py) and "
                               "the package has no MIOpen path; wrap a reference computation in layers.misc.library_reference()")
        return super().forward(x)


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
