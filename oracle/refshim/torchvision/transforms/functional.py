"""Stand-in for torchvision.transforms.functional (torchvision 0.8.2, absent here): the four calls the reference's
transforms make on PIL images, restated from torchvision's documented PIL back end.  Test infrastructure only."""
import numpy as np
import torch
from PIL import Image


def resize(img, size, interpolation=Image.BILINEAR):
    """size = (h, w) -> PIL resize to (w, h), bilinear (F.resize on a PIL image with a 2-sequence)."""
    if isinstance(size, int):
        w, h = img.size
        if (w <= h and w == size) or (h <= w and h == size):
            return img
        if w < h:
            return img.resize((size, int(size * h / w)), interpolation)
        return img.resize((int(size * w / h), size), interpolation)
    return img.resize(tuple(size[::-1]), interpolation)


def hflip(img):
    return img.transpose(Image.FLIP_LEFT_RIGHT)


def vflip(img):
    return img.transpose(Image.FLIP_TOP_BOTTOM)


def to_tensor(pic):
    """uint8 HWC -> float CHW in [0, 1] (division by 255 in fp32)."""
    arr = np.asarray(pic)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
    return t.float().div(255)


def normalize(tensor, mean, std, inplace=False):
    if not inplace:
        tensor = tensor.clone()
    mean = torch.as_tensor(mean, dtype=tensor.dtype)
    std = torch.as_tensor(std, dtype=tensor.dtype)
    tensor.sub_(mean[:, None, None]).div_(std[:, None, None])
    return tensor
