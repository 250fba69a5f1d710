"""CPU oracle of the data boundary -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(od_wscl_amd/) never does.

Restates, in numpy + Pillow:
  * Resize.get_size                       wetectron/data/transforms/transforms.py:41-61
  * the pixel chain Resize -> flips -> ToTensor -> Lighting -> Normalize      transforms.py:63-150
    (torchvision 0.8.2's PIL back end, a dependency absent from the reference tree and from this image:
     F.resize(img, (h, w)) = img.resize((w, h), PIL.Image.BILINEAR); F.hflip / F.vflip = img.transpose(...);
     F.to_tensor = uint8 HWC -> fp32 CHW / 255; F.normalize = (x - mean) / std)
  * to_image_list zero padding            structures/image_list.py:33-76
  * BoxList.resize / transpose / clip_to_image, remove_small_boxes, unique_boxes and the proposal preparation of
    PascalVOCDataset.__getitem__          structures/bounding_box.py:95-229, structures/boxlist_ops.py:96-113,
                                          data/datasets/coco.py:52-57, data/datasets/voc.py:94-111
  * `pil_bilinear_resize`: Pillow's 8-bit bilinear resampling itself (libImaging/Resample.c: precompute_coeffs,
    normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc), written out so that the HIP kernel has a
    line-by-line CPU twin; tests pin it against the installed Pillow on random images (bit-exact).

Pinning: tests/golden/data_voc.npz holds outputs of the imported reference's own transform chain, BatchCollator and
VOC proposal preparation (tests/golden/make_golden.py: gen_data); tests/test_oracle_vs_reference.py re-runs the live
reference when /root/reference is present.  torchvision being absent, the reference is imported with the functional
shim oracle/refshim/torchvision/transforms/functional.py, which restates the four F.* calls above; what the goldens
pin is therefore the reference's own logic (sizes, order, RNG consumption, box geometry, BGR/mean/std handling,
padding) on top of the installed Pillow's resampling.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def get_size(image_size, size, max_size):
    """transforms.py:41-61 with the drawn `size` given."""
    w, h = image_size
    if max_size is not None:
        lo, hi = float(min(w, h)), float(max(w, h))
        if hi / lo * size > max_size:
            size = int(round(max_size * lo / hi))
    if (w <= h and w == size) or (h <= w and h == size):
        return (h, w)
    if w < h:
        return (int(size * h / w), size)
    return (size, int(size * w / h))


def _coeffs(in_size, out_size):
    """precompute_coeffs (bilinear, box = whole axis) + normalize_coeffs_8bpc."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            v = 1.0 - t if t < 1.0 else 0.0
            w.append(v)
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One 8-bit resampling pass along `axis` (0 = vertical, 1 = horizontal) of an (H,W,C) uint8 array."""
    src = img.astype(np.int64)
    if axis == 1:
        src = src.transpose(1, 0, 2)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.int64)
    for xx in range(bounds.shape[0]):
        lo, n = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(n):
            acc += src[lo + t] * kk[xx, t]
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    if axis == 1:
        out = out.transpose(1, 0, 2)
    return out.astype(np.uint8)


def pil_bilinear_resize(img, out_h, out_w):
    """ImagingResample for an 8-bit image: horizontal pass first (rounded to uint8), then vertical; a pass is skipped
    when its size does not change."""
    img = np.ascontiguousarray(img, np.uint8)
    in_h, in_w = img.shape[:2]
    if out_w != in_w:
        img = _pass(img, *_coeffs(in_w, out_w), axis=1)
    if out_h != in_h:
        img = _pass(img, *_coeffs(in_h, out_h), axis=0)
    return img


def pixel_chain(pixels, out_hw, hflip, vflip, lighting, mean, std, to_bgr255, use_pillow=True):
    """uint8 (H,W,3) RGB -> fp32 (3,h,w): the reference's Resize -> flips -> ToTensor -> Lighting -> Normalize."""
    pixels = np.ascontiguousarray(pixels, np.uint8)
    oh, ow = out_hw
    if use_pillow:
        from PIL import Image
        img = Image.fromarray(pixels, "RGB")
        if (ow, oh) != img.size:
            img = img.resize((ow, oh), Image.BILINEAR)
        if hflip:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        if vflip:
            img = img.transpose(Image.FLIP_TOP_BOTTOM)
        arr = np.asarray(img)
    else:
        arr = pil_bilinear_resize(pixels, oh, ow)
        if hflip:
            arr = arr[:, ::-1]
        if vflip:
            arr = arr[::-1]
    x = arr.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    if lighting is not None:
        x = x + np.asarray(lighting, np.float32).reshape(3, 1, 1)
    if to_bgr255:
        x = x[[2, 1, 0]] * np.float32(255.0)
    x = (x - np.asarray(mean, np.float32).reshape(3, 1, 1)) / np.asarray(std, np.float32).reshape(3, 1, 1)
    return x.astype(np.float32)


def to_image_list(images, size_divisible=0):
    """structures/image_list.py:53-72 -> (batch (B,3,Hp,Wp) fp32, [(h, w)])."""
    h = max(im.shape[1] for im in images)
    w = max(im.shape[2] for im in images)
    if size_divisible > 0:
        h = int(math.ceil(h / size_divisible) * size_divisible)
        w = int(math.ceil(w / size_divisible) * size_divisible)
    batch = np.zeros((len(images), 3, h, w), np.float32)
    for im, slot in zip(images, batch):
        slot[:, : im.shape[1], : im.shape[2]] = im
    return batch, [tuple(im.shape[1:]) for im in images]


# ---- boxes ------------------------------------------------------------------------------------------------------
def boxes_resize(boxes, old_size, new_size):
    """bounding_box.py:95-131 on an (n,4) xyxy fp32 array; sizes are (w, h)."""
    boxes = np.asarray(boxes, np.float32)
    rw, rh = (float(s) / float(o) for s, o in zip(new_size, old_size))
    if rw == rh:
        return boxes * np.float32(rw)
    return boxes * np.asarray([rw, rh, rw, rh], np.float32)


def boxes_transpose(boxes, size, method):
    """bounding_box.py:133-169; method 0 = FLIP_LEFT_RIGHT (with the -1), 1 = FLIP_TOP_BOTTOM (without)."""
    boxes = np.asarray(boxes, np.float32)
    w, h = size
    out = boxes.copy()
    if method == 0:
        out[:, 0] = np.float32(w) - boxes[:, 2] - np.float32(1)
        out[:, 2] = np.float32(w) - boxes[:, 0] - np.float32(1)
    else:
        out[:, 1] = np.float32(h) - boxes[:, 3]
        out[:, 3] = np.float32(h) - boxes[:, 1]
    return out


def unique_boxes(boxes, scale=1.0):
    """data/datasets/coco.py:52-57: first occurrence of every distinct (rounded) box, in index order."""
    v = np.array([1, 1e3, 1e6, 1e9])
    hashes = np.round(boxes * scale).dot(v)
    _, index = np.unique(hashes, return_index=True)
    return np.sort(index)


def prepare_proposals(raw_boxes, image_size, min_size=20):
    """data/datasets/voc.py:94-111: de-duplicate, clip to the image (dropping empty boxes), drop boxes with a side
    (+1 convention) below 20 px.  raw_boxes = the int16 (n,4) array of the proposal file; image_size = (w, h)."""
    raw_boxes = np.asarray(raw_boxes)
    b = raw_boxes[unique_boxes(raw_boxes)].astype(np.float64).astype(np.float32)
    w, h = image_size
    b[:, 0] = np.clip(b[:, 0], 0, w - 1)
    b[:, 1] = np.clip(b[:, 1], 0, h - 1)
    b[:, 2] = np.clip(b[:, 2], 0, w - 1)
    b[:, 3] = np.clip(b[:, 3], 0, h - 1)
    b = b[(b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])]
    ws, hs = b[:, 2] - b[:, 0] + 1, b[:, 3] - b[:, 1] + 1
    return b[(ws >= min_size) & (hs >= min_size)]
