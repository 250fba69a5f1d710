import torch

from . import functional  # noqa: F401


class ColorJitter(object):
    """torchvision 0.8.2 draws `torch.randperm(4)` per call whatever the ranges; with every range empty (the only
    case the goldens use) the image is returned unchanged."""

    def __init__(self, brightness=None, contrast=None, saturation=None, hue=None):
        if any(v not in (None, 0, 0.0) for v in (brightness, contrast, saturation, hue)):
            raise NotImplementedError("shim: only the zero-jitter configuration is modelled")

    def __call__(self, x):
        torch.randperm(4)
        return x
