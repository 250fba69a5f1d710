"""Import the read-only reference (/root/reference) in THIS container.

TEST INFRASTRUCTURE ONLY -- used by tests/golden/make_golden.py to generate the
committed golden vectors and by `-m "not gpu"` tests that re-validate the oracle
against the live reference when the tree is present.  The reference Python
never travels to the GPU box; only the arrays it produced do.

What is patched (none of it is reference code):
  * stub packages for modules absent from the image (oracle/refshim/): yacs,
    apex (amp == identity at O0), torchvision (ops.nms restated), cv2, fvcore,
    IPython, pycocotools, tensorboardX
  * torch._six, torch.hub._download_url_to_file (removed from modern torch;
    wetectron/utils/imports.py:8, utils/model_zoo.py:10)
  * torch.Tensor.cuda -> identity (roi_heads/sim_head/sim_loss.py:72 calls
    .cuda() unconditionally)
  * wetectron._C -> the reference's own compiled CPU extension (oracle/_ref)
    with roi_pool_forward/backward + roi_align_backward supplied by the C
    oracle (the reference has no CPU implementation of those: csrc/ROIPool.h:23)
"""
import ctypes
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"
_STATE = {}


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "wetectron"))


def oracle_lib():
    """ctypes handle on oracle/liboracle.so (built by `make -C oracle`)."""
    if "lib" not in _STATE:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
        _STATE["lib"] = ctypes.CDLL(path)
    return _STATE["lib"]


def _fp(t):
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))


def _ip(t):
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_int32))


def c_roi_pool_forward(inp, rois, scale, ph, pw):
    lib = oracle_lib()
    inp = inp.contiguous().float()
    rois = rois.contiguous().float()
    B, C, H, W = inp.shape
    R = rois.shape[0]
    out = torch.empty(R, C, ph, pw, dtype=torch.float32)
    arg = torch.zeros(R, C, ph, pw, dtype=torch.int32)
    if out.numel():
        lib.oracle_roi_pool_fwd(_fp(inp), _fp(rois), ctypes.c_float(scale), B, C, H, W, R, ph, pw,
                                _fp(out), _ip(arg))
    return out, arg


def c_roi_pool_backward(grad, inp, rois, argmax, scale, ph, pw, B, C, H, W):
    lib = oracle_lib()
    grad = grad.contiguous().float()
    rois = rois.contiguous().float()
    gin = torch.zeros(B, C, H, W, dtype=torch.float32)
    if grad.numel():
        lib.oracle_roi_pool_bwd(_fp(grad), _ip(argmax.contiguous()), _fp(rois), B, C, H, W,
                                rois.shape[0], ph, pw, _fp(gin))
    return gin


def c_roi_align_forward(inp, rois, scale, ph, pw, sr):
    lib = oracle_lib()
    inp = inp.contiguous().float()
    rois = rois.contiguous().float()
    B, C, H, W = inp.shape
    R = rois.shape[0]
    out = torch.empty(R, C, ph, pw, dtype=torch.float32)
    if out.numel():
        lib.oracle_roi_align_fwd(_fp(inp), _fp(rois), ctypes.c_float(scale), B, C, H, W, R, ph, pw,
                                 int(sr), _fp(out))
    return out


def c_roi_align_backward(grad, rois, scale, ph, pw, B, C, H, W, sr):
    lib = oracle_lib()
    grad = grad.contiguous().float()
    rois = rois.contiguous().float()
    gin = torch.zeros(B, C, H, W, dtype=torch.float32)
    if grad.numel():
        lib.oracle_roi_align_bwd(_fp(grad), _fp(rois), ctypes.c_float(scale), B, C, H, W,
                                 rois.shape[0], ph, pw, int(sr), _fp(gin))
    return gin


def load_reference():
    """Returns the imported `wetectron` package (cached)."""
    if "wetectron" in _STATE:
        return _STATE["wetectron"]
    if not reference_available():
        raise RuntimeError("reference tree not present (expected at %s)" % REF_ROOT)
    sys.dont_write_bytecode = True  # the reference tree is read-only
    shim = os.path.join(HERE, "refshim")
    for p in (REF_ROOT, shim):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, shim)

    torch._six = types.SimpleNamespace(PY3=True, string_classes=(str,))
    if not hasattr(torch.hub, "_download_url_to_file"):
        torch.hub._download_url_to_file = torch.hub.download_url_to_file
    torch.Tensor.cuda = lambda self, *a, **k: self

    sys.path.insert(0, HERE)
    import build_ref
    build_ref.build()
    ref_c = build_ref.load_ref()

    ns = types.ModuleType("wetectron._C")
    ns.nms = ref_c.nms
    ns.roi_align_forward = ref_c.roi_align_forward
    ns.roi_align_backward = c_roi_align_backward
    ns.roi_pool_forward = c_roi_pool_forward
    ns.roi_pool_backward = c_roi_pool_backward
    ns._ref = ref_c
    sys.modules["wetectron._C"] = ns
    import wetectron
    wetectron._C = ns
    _STATE["wetectron"] = wetectron
    _STATE["ref_c"] = ref_c
    return wetectron


def reference_cfg(yaml_rel="configs/voc/voc07_contra_db_b8_lr0.01_mcg.yaml", opts=()):
    """A fresh reference cfg merged from one of its shipped yaml files."""
    load_reference()
    from wetectron.config import cfg as global_cfg
    global_cfg.defrost()
    # the reference reads ONE process-wide cfg (modules import it): restore its defaults first, so that a case built after
    # another one (R-50 yaml, then the COCO yaml) does not inherit the earlier case's keys
    if "cfg_defaults" not in _STATE:
        _STATE["cfg_defaults"] = global_cfg.clone()
    fresh = _STATE["cfg_defaults"].clone()
    for k in list(global_cfg.keys()):
        dict.__delitem__(global_cfg, k)
    for k, v in fresh.items():
        dict.__setitem__(global_cfg, k, v)
    global_cfg.merge_from_file(os.path.join(REF_ROOT, yaml_rel))
    global_cfg.merge_from_list(["MODEL.DEVICE", "cpu"] + list(opts))
    return global_cfg


def build_reference_model(cfg):
    load_reference()
    from wetectron.modeling.detector import build_detection_model
    return build_detection_model(cfg)


def to_np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
