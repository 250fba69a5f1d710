"""Data boundary timing on one MI355X: decoded uint8 image -> normalised padded fp32 batch slot.
GPU: pinned upload + csrc/preprocess.hip (3 launches); CPU: the reference's chain restated by the oracle (Pillow resize,
ToTensor, Normalize) + the fp32 upload the reference does instead."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _C
from od_wscl_amd.data import transforms as T
from od_wscl_amd.structures.image_list import to_image_list
from oracle import data_ref as D

dev = torch.device("cuda:0")
MEAN, STD = [102.9801, 115.9465, 122.7717], [1.0, 1.0, 1.0]
rng = np.random.default_rng(0)
for (h, w), size, mx in (((375, 500), 600, 2000), ((375, 500), 1200, 2000), ((333, 500), 480, 2000)):
    pixels = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    oh, ow = D.get_size((w, h), size, mx)
    Hp, Wp = -(-oh // 32) * 32, -(-ow // 32) * 32
    out = torch.empty((3, Hp, Wp), device=dev)
    px = torch.from_numpy(pixels).to(dev)
    for _ in range(3):
        _C.image_preprocess(px, (oh, ow), out, MEAN, STD)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        _C.image_preprocess(px, (oh, ow), out, MEAN, STD)
    e1.record(); torch.cuda.synchronize()
    k_us = e0.elapsed_time(e1) / n * 1e3
    # whole boundary: transform plan + staged upload + kernel, wall clock
    tr = T.Compose([T.Resize(size, mx), T.ToTensor(), T.Normalize(MEAN, STD)])
    for _ in range(3):
        to_image_list([tr(T.DeferredImage(pixels))[0]], 32).to(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        to_image_list([tr(T.DeferredImage(pixels))[0]], 32).to(dev)
    torch.cuda.synchronize(); g_ms = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    m = 5
    for _ in range(m):
        x = D.pixel_chain(pixels, (oh, ow), False, False, None, MEAN, STD, True)
        b, _ = D.to_image_list([x], 32)
        torch.from_numpy(b).to(dev)
    torch.cuda.synchronize(); c_ms = (time.perf_counter() - t0) / m * 1e3
    algo = h * w * 3 + 2 * h * ow * 3 + 3 * Hp * Wp * 4
    print("%dx%d -> %dx%d (pad %dx%d): kernels %.1f us = %.0f GB/s of %.1f MB algorithmic; boundary GPU path %.3f ms/img, "
          "CPU chain + fp32 upload %.2f ms/img (%.0fx)" % (h, w, oh, ow, Hp, Wp, k_us, algo / k_us / 1e3, algo / 1e6,
                                                          g_ms, c_ms, c_ms / g_ms))
