"""Deterministic synthetic inputs shaped like the reference's data pipeline output.

There are no datasets or checkpoints on the build/bench machines, so images,
proposal boxes, image-level labels and weights are pure functions of a seed
(counter-based generator, utils/rng.py) -- the CPU checker and the GPU path
regenerate identical tensors instead of shipping them.

Shapes follow what `make_data_loader` hands to `do_train`
(wetectron/data/collate_batch.py:15-25, engine/trainer.py:79-99):
  images  : fp32 (B,3,H,W) BGR, 0..255 minus PIXEL_MEAN (config/defaults.py:66),
            zero padded to a multiple of SIZE_DIVISIBILITY=32
  rois    : per image (P,4) xyxy integer-valued boxes inside the resized image,
            w,h >= 20 px (cfg.min_size), unique (data/datasets/voc.py:101-111)
  targets : per image 1..3 foreground labels in 1..C-1
"""
import math

import numpy as np

from .utils import rng

PIXEL_MEAN = (102.9801, 115.9465, 122.7717)  # config/defaults.py:66

# stream ids reserved for synthetic data (model RNG streams start at 1 << 20)
_S_IMAGE, _S_BOX, _S_LABEL, _S_WEIGHT = 100, 200, 300, 1000


def pad_to(x, div=32):
    return int(math.ceil(x / div) * div)


def make_image(seed, index, height, width, div=32):
    """(3,Hp,Wp) float32: U(0,255) - mean inside (height,width), zero padding outside."""
    hp, wp = pad_to(height, div), pad_to(width, div)
    img = np.zeros((3, hp, wp), np.float32)
    u = rng.uniform(seed, _S_IMAGE + index, 3 * height * width).reshape(3, height, width)
    for c in range(3):
        img[c, :height, :width] = u[c] * np.float32(255.0) - np.float32(PIXEL_MEAN[c])
    return img


def make_proposals(seed, index, n, height, width, min_size=20, n_objects=3, frac_cluster=0.3):
    """(n,4) float32 integer-valued xyxy boxes: 70 % log-uniform sides anywhere, 30 % jittered
    (+-15 %) around `n_objects` object boxes so IoU>=0.5 neighbourhoods have MCG-like sizes."""
    stream = _S_BOX + index
    ou = rng.uniform(seed, stream, 4 * n_objects).reshape(n_objects, 4)
    obj_w = (0.25 + 0.45 * ou[:, 0]) * width
    obj_h = (0.25 + 0.45 * ou[:, 1]) * height
    obj_x = ou[:, 2] * (width - obj_w)
    obj_y = ou[:, 3] * (height - obj_h)
    out, seen = [], set()
    chunk, t = 4096, 0
    lw = math.log(max(width / min_size, 1.0001))
    lh = math.log(max(height / min_size, 1.0001))
    while len(out) < n:
        if t > 256:
            raise ValueError("cannot draw %d unique proposals in a %dx%d image" % (n, width, height))
        u = rng.uniform(seed, stream, 6 * chunk, offset=4 * n_objects + t * 6 * chunk).reshape(chunk, 6)
        t += 1
        for k in range(chunk):
            if u[k, 5] < frac_cluster:
                g = int(u[k, 4] * n_objects) % n_objects
                jw = obj_w[g] * (0.85 + 0.3 * u[k, 0])
                jh = obj_h[g] * (0.85 + 0.3 * u[k, 1])
                x1 = obj_x[g] + (u[k, 2] - 0.5) * 0.3 * obj_w[g]
                y1 = obj_y[g] + (u[k, 3] - 0.5) * 0.3 * obj_h[g]
            else:
                jw = min_size * math.exp(u[k, 0] * lw)
                jh = min_size * math.exp(u[k, 1] * lh)
                x1 = u[k, 2] * max(width - jw, 0.0)
                y1 = u[k, 3] * max(height - jh, 0.0)
            x1 = int(min(max(round(float(x1)), 0), width - 1))
            y1 = int(min(max(round(float(y1)), 0), height - 1))
            x2 = int(min(x1 + round(float(jw)), width - 1))
            y2 = int(min(y1 + round(float(jh)), height - 1))
            if x2 - x1 < min_size or y2 - y1 < min_size:
                continue
            key = (x1, y1, x2, y2)
            if key in seen:
                continue
            seen.add(key)
            out.append(key)
            if len(out) == n:
                break
    return np.asarray(out, np.float32).reshape(n, 4)


def make_labels(seed, index, num_classes, max_labels=3):
    """sorted unique int64 foreground labels (1..C-1), 1..max_labels of them."""
    u = rng.uniform(seed, _S_LABEL + index, 1 + max_labels)
    k = 1 + int(u[0] * max_labels) % max_labels
    labs = sorted(set(1 + int(v * (num_classes - 1)) % (num_classes - 1) for v in u[1:1 + k]))
    return np.asarray(labs, np.int64)


def fill_weights(seed, stream, shape, std):
    """Zero-mean values of standard deviation `std`: sqrt(12) std (u - 1/2) with u the
    counter-based uniform stream (a uniform law, 6x cheaper to draw than Box-Muller for the
    153 M parameters; only the variance matters to the hot path)."""
    n = int(np.prod(shape))
    out = np.empty(n, np.float32)
    step = 1 << 22
    scale = np.float32(math.sqrt(12.0) * std)

    def chunk(off):
        m = min(step, n - off)
        out[off:off + m] = (rng.uniform(seed, stream, m, off) - np.float32(0.5)) * scale

    offs = list(range(0, n, step))
    if len(offs) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, len(offs))) as ex:
            list(ex.map(chunk, offs))
    else:
        for off in offs:
            chunk(off)
    return out.reshape(shape)


_S_BUFFER = 1 << 16


def init_buffers(named_shapes, seed):
    """Formula values for the frozen batch-norm buffers of the ResNet bodies (layers/batch_norm.py:12-17):
    weight in [0.5, 1) (x0.5 on the last BN of a block so the residual trunk stays O(1) over 16 blocks),
    bias and running_mean in [-0.1, 0.1), running_var in [0.5, 1.5) -- the spread a trained network has,
    so the affine is exercised, not the identity the constructor leaves."""
    sd = {}
    for pos, (name, shape) in enumerate(named_shapes):
        u = rng.uniform(seed, _S_BUFFER + pos, int(np.prod(shape)), 0).reshape(shape)
        if name.endswith("running_var"):
            v = np.float32(0.5) + u
        elif name.endswith(".weight"):
            v = (np.float32(0.5) + np.float32(0.5) * u) * np.float32(0.5 if ".bn3." in name else 1.0)
        else:
            v = (u - np.float32(0.5)) * np.float32(0.2)
        sd[name] = v.astype(np.float32)
    return sd


def init_state_dict(named_shapes, seed, scheme="reference", overrides=None):
    """Formula weights for every parameter of the detector.

    named_shapes: ordered [(name, shape)] with the reference's parameter names
    (SURVEY.md s5 checkpoint row).  scheme "reference" follows the reference's
    initialisers in VARIANCE (values are uniform, see fill_weights): conv kaiming fan_out (vgg16.py:38-41),
    fc6/fc7 N(0,0.01) (vgg16.py:142-146), predictor N(0,0.001)
    (roi_weak_predictors.py:136-140), Sim_Net kaiming-normal fan_out
    (sim_net.py:19-23), biases 0.  `overrides` maps a name substring to a std
    (parity fixtures use larger predictor weights so scores are well separated).
    Values are a function of (seed, parameter position), not of torch's RNG."""
    overrides = overrides or {}
    sd = {}
    for pos, (name, shape) in enumerate(named_shapes):
        std = None
        for key, val in overrides.items():
            if key in name:
                std = val
        if name.endswith(".bias") and std is None:
            sd[name] = np.zeros(shape, np.float32)
            continue
        if std is None:
            if "features" in name or "model_sim" in name:
                fan_out = shape[0] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[0]
                std = math.sqrt(2.0 / fan_out)
            elif "body.stem" in name or "body.layer" in name:
                std = 1.0 / math.sqrt(int(np.prod(shape[1:])))     # kaiming_uniform_(a=1) (resnet.py:289,333,342,396)
            elif "classifier" in name:
                std = 0.01
            else:
                std = 0.001
        sd[name] = fill_weights(seed, _S_WEIGHT + pos, shape, std)
    return sd
