"""numpy front-end of oracle/liboracle.so (odw_oracle.c) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    if not os.path.exists(_PATH) or os.path.getmtime(_PATH) < os.path.getmtime(os.path.join(_HERE, "odw_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _PATH


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_supcon_v2.restype = ctypes.c_double
        _lib.oracle_nms_wt.restype = ctypes.c_int
        _lib.oracle_nms_tv.restype = ctypes.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, ty=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def roi_pool_fwd(feat, rois, scale, ph, pw):
    feat, rois = _f(feat), _f(rois)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.empty((R, C, ph, pw), np.float32)
    arg = np.zeros((R, C, ph, pw), np.int32)
    if out.size:
        lib().oracle_roi_pool_fwd(_p(feat), _p(rois), ctypes.c_float(scale), B, C, H, W, R, ph, pw,
                                  _p(out), _p(arg, ctypes.c_int32))
    return out, arg


def roi_pool_bwd(grad, argmax, rois, shape, ph, pw):
    grad, rois = _f(grad), _f(rois)
    argmax = np.ascontiguousarray(argmax, np.int32)
    B, C, H, W = shape
    gin = np.zeros((B, C, H, W), np.float32)
    if grad.size:
        lib().oracle_roi_pool_bwd(_p(grad), _p(argmax, ctypes.c_int32), _p(rois), B, C, H, W, rois.shape[0],
                                  ph, pw, _p(gin))
    return gin


def roi_align_fwd(feat, rois, scale, ph, pw, sr):
    feat, rois = _f(feat), _f(rois)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.empty((R, C, ph, pw), np.float32)
    if out.size:
        lib().oracle_roi_align_fwd(_p(feat), _p(rois), ctypes.c_float(scale), B, C, H, W, R, ph, pw, int(sr),
                                   _p(out))
    return out


def roi_align_bwd(grad, rois, scale, shape, ph, pw, sr):
    grad, rois = _f(grad), _f(rois)
    B, C, H, W = shape
    gin = np.zeros((B, C, H, W), np.float32)
    if grad.size:
        lib().oracle_roi_align_bwd(_p(grad), _p(rois), ctypes.c_float(scale), B, C, H, W, rois.shape[0], ph, pw,
                                   int(sr), _p(gin))
    return gin


def nms_wt(boxes, scores, thr, use_ge=True):
    boxes, scores = _f(boxes), _f(scores)
    n = boxes.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    k = lib().oracle_nms_wt(_p(boxes), _p(scores), n, ctypes.c_float(thr), int(bool(use_ge)),
                            _p(keep, ctypes.c_int64))
    return keep[:k].copy()


def nms_tv(boxes, scores, thr):
    boxes, scores = _f(boxes), _f(scores)
    n = boxes.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    k = lib().oracle_nms_tv(_p(boxes), _p(scores), n, ctypes.c_float(thr), _p(keep, ctypes.c_int64))
    return keep[:k].copy()


def box_iou(a, b):
    a, b = _f(a), _f(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    if out.size:
        lib().oracle_box_iou(_p(a), a.shape[0], _p(b), b.shape[0], _p(out))
    return out


def pairwise_sim(E):
    E = _f(E)
    P, D = E.shape
    S = np.empty((P, P), np.float32)
    if P:
        lib().oracle_pairwise_sim(_p(E), P, D, _p(S))
    return S


def supcon_v2(F, labels, w, tau, need_grad=True):
    F, w = _f(F), _f(w)
    labels = np.ascontiguousarray(labels, np.int32)
    N, D = F.shape
    dF = np.empty_like(F) if need_grad else None
    loss = lib().oracle_supcon_v2(_p(F), _p(labels, ctypes.c_int32), _p(w), N, D, ctypes.c_float(tau),
                                  _p(dF) if need_grad else None)
    return float(loss), dF
