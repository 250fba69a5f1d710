"""The transform chain of the data boundary (wetectron/data/transforms/transforms.py:17-150, build.py:18-73), split
the MI355X way: the GEOMETRY (which size, which flips, the boxes of targets and proposals) is decided and applied on
the host exactly like the reference -- same classes, same arguments, same draws from `random` / `torch` in the same
order -- while the PIXELS stay the decoded uint8 image plus a recorded plan (`DeferredImage`).  The plan is executed
by one fused gfx950 kernel when the batch is moved to the device (`PendingImageList.to`, csrc/preprocess.hip):
resize (bit-identical to Pillow's bilinear), flips, ToTensor, Lighting, Normalize and the zero padding of
`to_image_list` in one pass, from a 4x smaller host->device copy.  There is no host pixel path: moving a pending
batch anywhere but a GPU raises."""
import random

import numpy as np
import torch

from ..structures.bounding_box import FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM


class DeferredImage(object):
    """Decoded pixels (uint8, H x W x 3, RGB) + the pixel plan the transforms recorded."""

    def __init__(self, pixels):
        pixels = np.asarray(pixels)
        if pixels.dtype != np.uint8 or pixels.ndim != 3 or pixels.shape[2] != 3:
            raise ValueError("expected an RGB uint8 image (H,W,3), got %s %s" % (pixels.dtype, pixels.shape))
        self.pixels = np.ascontiguousarray(pixels)
        self.out_hw = None          # Resize target (h, w); None = native size
        self.hflip = False          # flips applied AFTER the resize (the reference's order)
        self.vflip = False
        self.tensor = False         # ToTensor seen
        self.light = None           # Lighting offset per RGB channel (fp32[3])
        self.norm = None            # (mean[3], std[3], to_bgr255)
        self.device_pixels = {}     # device -> uploaded copy of `pixels` (shared by forks: one upload, many plans)

    def fork(self):
        """A fresh plan over the same pixels (test-time augmentation runs many plans per image)."""
        if self.out_hw is not None or self.hflip or self.vflip or self.tensor:
            raise RuntimeError("fork() is for an image no transform has touched yet")
        twin = DeferredImage.__new__(DeferredImage)
        twin.pixels, twin.device_pixels = self.pixels, self.device_pixels
        twin.out_hw, twin.hflip, twin.vflip, twin.tensor, twin.light, twin.norm = None, False, False, False, None, None
        return twin

    @property
    def size(self):
        """(width, height) like PIL.Image.size -- of the image the plan produces."""
        h, w = self.out_hw if self.out_hw is not None else self.pixels.shape[:2]
        return (w, h)

    @property
    def shape(self):
        """(3, H, W): what to_image_list reads from a tensor."""
        w, h = self.size
        return (3, h, w)

    def _eager_only(self, what):
        if self.out_hw is not None or self.hflip or self.vflip or self.tensor:
            raise NotImplementedError("%s has to come before Resize / flips / ToTensor in the chain" % what)


def defer(image):
    """PIL image / ndarray / DeferredImage -> DeferredImage."""
    if isinstance(image, DeferredImage):
        return image
    if hasattr(image, "convert") and hasattr(image, "size"):      # PIL
        image = np.asarray(image.convert("RGB"))
    return DeferredImage(image)


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target=None, rois=None):
        image = defer(image)
        for t in self.transforms:
            image, target, rois = t(image, target, rois)
        return image, target, rois

    def __repr__(self):
        return self.__class__.__name__ + "(" + "".join("\n    {0}".format(t) for t in self.transforms) + "\n)"


class Resize(object):
    def __init__(self, min_size, max_size):
        if not isinstance(min_size, (list, tuple)):
            min_size = (min_size,)
        self.min_size = min_size
        self.max_size = max_size

    def get_size(self, image_size):
        """(h, w) after resizing: the short side becomes a drawn min_size, lowered when the long side would pass
        max_size; the long side is truncated to an integer (the reference's rule, transforms.py:41-61, with
        `random.choice` consumed exactly once)."""
        w, h = image_size
        short, long_ = (w, h) if w <= h else (h, w)
        target = random.choice(self.min_size)
        if self.max_size is not None and float(long_) / float(short) * target > self.max_size:
            target = int(round(self.max_size * float(short) / float(long_)))
        if short == target:
            return (h, w)
        other = int(target * long_ / short)
        return (other, target) if w < h else (target, other)

    def __call__(self, image, target=None, rois=None):
        image = defer(image)
        if image.out_hw is not None or image.hflip or image.vflip or image.tensor:
            raise NotImplementedError("one Resize per chain, before the flips and ToTensor")
        image.out_hw = tuple(self.get_size(image.size))
        if target is not None:
            target = target.resize(image.size)
        if rois is not None:
            rois = rois.resize(image.size)
        return image, target, rois


class RandomHorizontalFlip(object):
    def __init__(self, prob=0.5):
        self.prob = prob

    def __call__(self, image, target=None, rois=None):
        image = defer(image)
        if random.random() < self.prob:
            if image.tensor:
                raise NotImplementedError("flips come before ToTensor")
            image.hflip = not image.hflip
            if target is not None:
                target = target.transpose(FLIP_LEFT_RIGHT)
            if rois is not None:
                rois = rois.transpose(FLIP_LEFT_RIGHT)
        return image, target, rois


class RandomVerticalFlip(object):
    def __init__(self, prob=0.5):
        self.prob = prob

    def __call__(self, image, target=None, rois=None):
        image = defer(image)
        if random.random() < self.prob:
            if image.tensor:
                raise NotImplementedError("flips come before ToTensor")
            image.vflip = not image.vflip
            if target is not None:
                target = target.transpose(FLIP_TOP_BOTTOM)
            if rois is not None:
                rois = rois.transpose(FLIP_TOP_BOTTOM)
        return image, target, rois


class ColorJitter(object):
    """torchvision 0.8.2 ColorJitter on the PIL image (transforms.py:100-114).  It draws `torch.randperm(4)` per image
    even when every range is empty -- kept, so that the Lighting draw that follows sees the same generator state.
    Non-zero jitter is applied eagerly with PIL's ImageEnhance (brightness / contrast / saturation) and an HSV shift
    (hue), the operations torchvision's PIL back end runs; no shipped config enables it."""

    def __init__(self, brightness=None, contrast=None, saturation=None, hue=None):
        self.brightness = self._range(brightness, 1.0)
        self.contrast = self._range(contrast, 1.0)
        self.saturation = self._range(saturation, 1.0)
        self.hue = self._range(hue, 0.0, clip_first=False)

    @staticmethod
    def _range(value, center, clip_first=True):
        if value is None:
            return None
        if isinstance(value, (int, float)):
            if value < 0:
                raise ValueError("jitter strength has to be non negative")
            value = [center - float(value), center + float(value)]
            if clip_first:
                value[0] = max(value[0], 0.0)
        if value[0] == value[1] == center:
            return None
        return tuple(value)

    def __call__(self, image, target=None, rois=None):
        image = defer(image)
        order = torch.randperm(4)
        for fn_id in order.tolist():
            rng = (self.brightness, self.contrast, self.saturation, self.hue)[fn_id]
            if rng is None:
                continue
            factor = torch.tensor(1.0).uniform_(rng[0], rng[1]).item()
            image._eager_only("ColorJitter")
            image.pixels = _jitter(image.pixels, fn_id, factor)
            image.device_pixels = {}
        return image, target, rois


def _jitter(pixels, fn_id, factor):
    from PIL import Image, ImageEnhance
    img = Image.fromarray(pixels, "RGB")
    if fn_id == 0:
        img = ImageEnhance.Brightness(img).enhance(factor)
    elif fn_id == 1:
        img = ImageEnhance.Contrast(img).enhance(factor)
    elif fn_id == 2:
        img = ImageEnhance.Color(img).enhance(factor)
    else:
        h, s, v = img.convert("HSV").split()
        np_h = np.array(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            np_h += np.uint8(factor * 255)
        img = Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")
    return np.ascontiguousarray(np.asarray(img))


class ToTensor(object):
    def __call__(self, image, target=None, rois=None):
        image = defer(image)
        image.tensor = True
        return image, target, rois


class Normalize(object):
    def __init__(self, mean, std, to_bgr255=True):
        self.mean = mean
        self.std = std
        self.to_bgr255 = to_bgr255

    def __call__(self, image, target=None, rois=None):
        image = defer(image)
        if not image.tensor or image.norm is not None:
            raise NotImplementedError("Normalize follows ToTensor, once")
        image.norm = (np.asarray(self.mean, np.float32), np.asarray(self.std, np.float32), bool(self.to_bgr255))
        return image, target, rois


class Lighting(object):
    """AlexNet-style PCA lighting noise (transforms.py:133-150): one fp32 offset per RGB channel, added after
    ToTensor.  The three normal draws come from torch's global generator like the reference's
    `img.new().resize_(3).normal_(0, alphastd)`; the offset rides in the pixel plan."""

    def __init__(self, alphastd, eigval, eigvec):
        self.alphastd = alphastd
        self.eigval = eigval
        self.eigvec = eigvec

    def __call__(self, img, target=None, rois=None):
        img = defer(img)
        if self.alphastd == 0:
            return img, target, rois
        if not img.tensor or img.norm is not None or img.light is not None:
            raise NotImplementedError("Lighting sits between ToTensor and Normalize, once")
        alpha = torch.empty(3, dtype=torch.float32).normal_(0, self.alphastd)
        rgb = self.eigvec.float().clone().mul(alpha.view(1, 3).expand(3, 3)) \
            .mul(self.eigval.float().view(1, 3).expand(3, 3)).sum(1).squeeze()
        img.light = rgb.numpy().astype(np.float32)
        return img, target, rois


_imagenet_pca = {
    "eigval": torch.Tensor([0.2175, 0.0188, 0.0045]),
    "eigvec": torch.Tensor([
        [-0.5675, 0.7192, 0.4009],
        [-0.5808, -0.0045, -0.8140],
        [-0.5836, -0.6948, 0.4203],
    ]),
}


def build_transforms(cfg, is_train=True):
    """data/transforms/build.py:18-73: jitter, Resize, flips (horizontal 0.5 in training whatever the config says,
    :22), ToTensor, Lighting(0.1) when INPUT.PCA, Normalize."""
    if is_train:
        min_size, max_size = cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN
        flip_h, flip_v = 0.5, cfg.INPUT.VERTICAL_FLIP_PROB_TRAIN
        jitter = (cfg.INPUT.BRIGHTNESS, cfg.INPUT.CONTRAST, cfg.INPUT.SATURATION, cfg.INPUT.HUE)
    else:
        min_size, max_size = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST
        flip_h, flip_v = 0.0, 0.0
        jitter = (0.0, 0.0, 0.0, 0.0)
    chain = [ColorJitter(*jitter), Resize(min_size, max_size), RandomHorizontalFlip(flip_h),
             RandomVerticalFlip(flip_v), ToTensor()]
    if cfg.INPUT.PCA:
        chain.append(Lighting(0.1, _imagenet_pca["eigval"], _imagenet_pca["eigvec"]))
    chain.append(Normalize(mean=cfg.INPUT.PIXEL_MEAN, std=cfg.INPUT.PIXEL_STD, to_bgr255=cfg.INPUT.TO_BGR255))
    return Compose(chain)
