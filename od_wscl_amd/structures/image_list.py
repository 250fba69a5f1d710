"""ImageList / to_image_list (wetectron/structures/image_list.py:11-76): images of
different sizes zero-padded into one (B,3,H,W) tensor whose H,W are multiples of
`size_divisible`."""
import math

import numpy as np
import torch


class ImageList(object):
    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


class _PinnedRing(object):
    """Pinned uint8 staging buffers for the host->device copies of decoded images: a buffer is reused once the copy
    that read it has completed (its event), so `.to(device)` never blocks on the previous batch."""

    def __init__(self):
        self.slots = []      # [buffer, event]

    def stage(self, array, device):
        n = array.size
        slot = None
        for s in self.slots:
            if s[0].numel() >= n and s[1].query():
                slot = s
                break
        if slot is None:
            slot = [torch.empty(max(n, 1 << 20), dtype=torch.uint8).pin_memory(), torch.cuda.Event()]
            self.slots.append(slot)
        slot[0][:n].numpy()[...] = array.reshape(-1)
        dev = slot[0][:n].to(device, non_blocking=True)
        slot[1].record()
        return dev.view(array.shape)


_RING = _PinnedRing()


class PendingImageList(object):
    """What BatchCollator returns for images that still are decoded pixels + a plan (data/transforms.DeferredImage):
    it knows every image size and the padded batch shape, and `.to(gpu)` runs the plans on the device -- one uint8
    copy and one fused preprocessing call per image, written straight into the padded batch (csrc/preprocess.hip).
    Mirrors ImageList's interface (`image_sizes`, `.to`), so the trainer's `images.to(device)` is the trigger."""

    def __init__(self, images, size_divisible=0):
        self.images = list(images)
        self.image_sizes = [torch.Size(im.shape[-2:]) for im in self.images]
        h = max(s[0] for s in self.image_sizes)
        w = max(s[1] for s in self.image_sizes)
        if size_divisible > 0:
            h = int(math.ceil(h / size_divisible) * size_divisible)
            w = int(math.ceil(w / size_divisible) * size_divisible)
        self.padded_hw = (h, w)

    def __len__(self):
        return len(self.images)

    def to(self, device, *args, **kwargs):
        from .. import _C
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("od_wscl_amd: a pending image batch is materialised by the GPU preprocessing kernel "
                               "only -- there is no host pixel path")
        with torch.cuda.device(device):
            batch = torch.empty((len(self.images), 3) + self.padded_hw, dtype=torch.float32, device=device)
            for im, slot in zip(self.images, batch):
                if not im.tensor or im.norm is None:
                    raise RuntimeError("the transform chain has to end with ToTensor + Normalize")
                mean, std, bgr = im.norm
                pixels = im.device_pixels.get(device)
                if pixels is None:
                    pixels = im.device_pixels[device] = _RING.stage(im.pixels, device)
                _C.image_preprocess(pixels, im.shape[-2:], slot, mean, std, bgr,
                                    hflip=im.hflip, vflip=im.vflip, lighting=im.light)
        return ImageList(batch, self.image_sizes)


def to_image_list(tensors, size_divisible=0):
    if isinstance(tensors, (ImageList, PendingImageList)):
        return tensors
    if isinstance(tensors, (tuple, list)) and len(tensors) and hasattr(tensors[0], "pixels"):
        return PendingImageList(tensors, size_divisible)
    if hasattr(tensors, "pixels"):
        return PendingImageList([tensors], size_divisible)
    if isinstance(tensors, torch.Tensor) and size_divisible > 0:
        tensors = [tensors]
    if isinstance(tensors, torch.Tensor):
        if tensors.dim() == 3:
            tensors = tensors[None]
        assert tensors.dim() == 4
        return ImageList(tensors, [t.shape[-2:] for t in tensors])
    if isinstance(tensors, (tuple, list)):
        c = tensors[0].shape[0]
        h = max(t.shape[1] for t in tensors)
        w = max(t.shape[2] for t in tensors)
        if size_divisible > 0:
            h = int(math.ceil(h / size_divisible) * size_divisible)
            w = int(math.ceil(w / size_divisible) * size_divisible)
        batch = tensors[0].new_zeros((len(tensors), c, h, w))
        for img, slot in zip(tensors, batch):
            slot[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        return ImageList(batch, [t.shape[-2:] for t in tensors])
    raise TypeError("Unsupported type for to_image_list: %s" % type(tensors))
