"""A SECOND, independent CPU restatement of the reference's ROIPool -- TEST INFRASTRUCTURE ONLY.

The reference ships no CPU ROIPool (wetectron/csrc/ROIPool.h:23: "Not implemented on the CPU"), so the C oracle
(oracle/odw_oracle.c) is itself a restatement of the CUDA kernel, and inside the imported reference it stands in for
`_C.roi_pool_forward` -- every end-to-end golden therefore contains the oracle's own pooling.  This file pins that
oracle from a second side: written in plain numpy directly from the behaviour of
wetectron/csrc/cuda/ROIPool_cuda.cu:17-77 (forward) and :80-108 (backward), without reference to odw_oracle.c, and
swept against both the C oracle and the HIP kernels over >= 10^4 random and adversarial ROIs
(tests/test_roi_pool_pin.py).

Behaviour restated (ROIPool_cuda.cu line numbers):
  :28-33  batch index = int(roi[0]); the four corners are ROUNDED products in single precision: the fp32 product
          coord * scale, then C round() = half AWAY from zero (not numpy's half-to-even), then int;
  :36-37  width / height = max(end - start + 1, 1): malformed (inverted) ROIs become 1 x 1;
  :38-41  bin sizes are fp32 quotients float(height) / float(PH);
  :43-50  hstart = floor(fp32(ph) * bin_h), hend = ceil(fp32(ph + 1) * bin_h), both fp32 products;
  :53-56  + roi start, clipped to [0, H] / [0, W];   :57 empty when hend <= hstart or wend <= wstart;
  :60-62  an empty bin gives 0 with argmax -1; otherwise the scan starts from -FLT_MAX with argmax -1;
  :65-73  row-major scan with a STRICT '>' : the first maximum wins; a window that holds nothing above -FLT_MAX
          keeps (-FLT_MAX, -1); NaNs never win a comparison;
  :75-76  outputs: the maximum and its FLAT position h * W + w inside the (H, W) plane (int32).
Backward (:92-104): grad_in[batch, c].flat[argmax] += grad_out wherever argmax != -1.
"""
import numpy as np

_F32_LOWEST = np.float32(-3.4028234663852886e38)        # -FLT_MAX


def _round_half_away(x):
    """C round() of float32 values, as int64."""
    x = np.asarray(x, np.float32).astype(np.float64)       # exact widening: the comparison below is then exact too
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)


def bin_edges(start, length, nbins, limit):
    """[lo, hi) of the `nbins` bins of one ROI axis, clipped to [0, limit] (ROIPool_cuda.cu:38-56)."""
    size = np.float32(length) / np.float32(nbins)
    idx = np.arange(nbins, dtype=np.float32)
    lo = np.floor(idx * size).astype(np.int64)                         # fp32 product, then floor
    hi = np.ceil((idx + np.float32(1.0)) * size).astype(np.int64)      # fp32 product, then ceil
    lo = np.minimum(np.maximum(lo + start, 0), limit)
    hi = np.minimum(np.maximum(hi + start, 0), limit)
    return lo, hi


def roi_pool_forward(feat, rois, scale, ph, pw):
    """feat (B, C, H, W) fp32, rois (R, 5) fp32 [batch, x1, y1, x2, y2] -> (out (R, C, ph, pw) fp32, argmax int32)."""
    feat = np.ascontiguousarray(feat, np.float32)
    rois = np.ascontiguousarray(rois, np.float32)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, ph, pw), np.float32)
    arg = np.full((R, C, ph, pw), -1, np.int32)
    s = np.float32(scale)
    corners = _round_half_away(rois[:, 1:5] * s)                       # fp32 products (both operands are fp32)
    for r in range(R):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = (int(v) for v in corners[r])
        rw, rh = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1)
        hlo, hhi = bin_edges(y1, rh, ph, H)
        wlo, whi = bin_edges(x1, rw, pw, W)
        plane = feat[b]                                                # (C, H, W)
        for i in range(ph):
            if hhi[i] <= hlo[i]:
                continue                                               # empty: 0 / -1 (the initial values)
            for j in range(pw):
                if whi[j] <= wlo[j]:
                    continue
                win = plane[:, hlo[i]:hhi[i], wlo[j]:whi[j]]
                ww = win.shape[2]
                flat = win.reshape(C, -1)
                cand = np.where(np.isnan(flat), -np.inf, flat)         # a NaN never satisfies '>'
                k = np.argmax(cand, axis=1)                            # first maximum in row-major order
                best = cand[np.arange(C), k]
                took = best > _F32_LOWEST                              # something beat the initial -FLT_MAX
                out[r, :, i, j] = np.where(took, best, _F32_LOWEST)
                pos = (hlo[i] + k // ww) * W + (wlo[j] + k % ww)
                arg[r, :, i, j] = np.where(took, pos, -1)
    return out, arg


def roi_pool_backward(grad, argmax, rois, shape):
    """grad (R, C, ph, pw), argmax int32 -> grad_in (B, C, H, W) float64 sums (exact reference for any fp32 order)."""
    B, C, H, W = shape
    gin = np.zeros((B, C, H * W), np.float64)
    R = grad.shape[0]
    g = np.asarray(grad, np.float64).reshape(R, C, -1)
    a = np.asarray(argmax).reshape(R, C, -1)
    cidx = np.broadcast_to(np.arange(C)[:, None], a.shape[1:])
    for r in range(R):
        b = int(rois[r, 0])
        ok = a[r] >= 0
        np.add.at(gin[b], (cidx[ok], a[r][ok]), g[r][ok])
    return gin.reshape(B, C, H, W)
