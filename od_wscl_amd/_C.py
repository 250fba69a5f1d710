"""`wetectron._C`-shaped module: the five operators the reference binds through
pybind11 (csrc/vision.cpp:9-25), same names, argument order and return values,
backed by libodwscl.so.  `wetectron/layers/*.py` works unchanged with
`from od_wscl_amd import _C`.
"""
import os

import torch

from . import _lib as L

_NMS_MODE = {"tv": 0, "wt_ge": 1, "wt_gt": 2}


def roi_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width):
    """-> (output (R,C,PH,PW) fp32, argmax (R,C,PH,PW) int32).  csrc/ROIPool.h:11-24."""
    L.need_gpu(input, rois)
    input = input.contiguous().float()   # ROIPool_cuda.cu:140 takes .contiguous()
    rois = rois.contiguous().float()
    B, C, H, W = input.shape
    R = rois.shape[0]
    out = torch.empty((R, C, pooled_height, pooled_width), dtype=torch.float32, device=input.device)
    argmax = torch.empty((R, C, pooled_height, pooled_width), dtype=torch.int32, device=input.device)     # every entry is written
    if out.numel() == 0:
        return out, argmax
    nbytes = L.lib().odw_roi_pool_forward_workspace(B, C, H, W, R, pooled_height, pooled_width)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=input.device)
    L.check(L.lib().odw_roi_pool_forward(L.ptr(input), L.ptr(rois), float(spatial_scale), B, C, H, W, R,
                                         pooled_height, pooled_width, L.ptr(out), L.ptr(argmax),
                                         L.ptr(ws), nbytes, L.stream()), "roi_pool_forward")
    return out, argmax


def roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_height, pooled_width,
                      batch_size, channels, height, width):
    """-> grad_input (B,C,H,W).  csrc/ROIPool.h:26-45 (input is only used for its sizes)."""
    L.need_gpu(grad, rois, argmax)
    grad = grad.contiguous().float()
    rois = rois.contiguous().float()
    argmax = argmax.contiguous()
    gin = torch.empty((batch_size, channels, height, width), dtype=torch.float32, device=grad.device)
    if gin.numel() == 0:
        return gin
    if os.environ.get("ODW_POOL_BWD_ATOMIC") == "1":       # float LDS atomics: order-dependent rounding (comparison)
        L.check(L.lib().odw_roi_pool_backward(L.ptr(grad), L.ptr(argmax), L.ptr(rois), batch_size, channels,
                                              height, width, rois.shape[0], pooled_height, pooled_width,
                                              L.ptr(gin), L.stream()), "roi_pool_backward")
        return gin
    ws = torch.empty(64, dtype=torch.uint8, device=grad.device)
    L.check(L.lib().odw_roi_pool_backward_det(L.ptr(grad), L.ptr(argmax), L.ptr(rois), batch_size, channels,
                                              height, width, rois.shape[0], pooled_height, pooled_width,
                                              L.ptr(gin), L.ptr(ws), 64, L.stream()), "roi_pool_backward_det")
    return gin


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    """-> output (R,C,PH,PW).  csrc/ROIAlign.h:11-25."""
    L.need_gpu(input, rois)
    input = input.contiguous().float()
    rois = rois.contiguous().float()
    B, C, H, W = input.shape
    R = rois.shape[0]
    out = torch.empty((R, C, pooled_height, pooled_width), dtype=torch.float32, device=input.device)
    if out.numel() == 0:
        return out
    ws_bytes = L.lib().odw_roi_align_forward_workspace(B, C, H, W, R, pooled_height, pooled_width)
    ws = torch.empty(max(int(ws_bytes), 1), dtype=torch.uint8, device=input.device)
    L.check(L.lib().odw_roi_align_forward_ws(L.ptr(input), L.ptr(rois), float(spatial_scale), B, C, H, W, R,
                                             pooled_height, pooled_width, int(sampling_ratio), L.ptr(out), L.ptr(ws),
                                             int(ws_bytes), L.stream()), "roi_align_forward")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                       height, width, sampling_ratio):
    """-> grad_input (B,C,H,W).  csrc/ROIAlign.h:27-45."""
    L.need_gpu(grad, rois)
    grad = grad.contiguous().float()
    rois = rois.contiguous().float()
    gin = torch.empty((batch_size, channels, height, width), dtype=torch.float32, device=grad.device)
    if gin.numel() == 0:
        return gin
    ws_bytes = L.lib().odw_roi_align_backward_workspace(rois.shape[0], pooled_height, pooled_width)
    ws = torch.empty(max(int(ws_bytes), 1), dtype=torch.uint8, device=grad.device)
    L.check(L.lib().odw_roi_align_backward_ws(L.ptr(grad), L.ptr(rois), float(spatial_scale), batch_size,
                                              channels, height, width, rois.shape[0], pooled_height,
                                              pooled_width, int(sampling_ratio), L.ptr(gin), L.ptr(ws), int(ws_bytes),
                                              L.stream()), "roi_align_backward")
    return gin


def _nms(dets, scores, threshold, mode):
    L.need_gpu(dets, scores)
    n = dets.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=dets.device)
    dets = dets.contiguous().float()
    scores = scores.contiguous().float()
    keep = torch.empty((n,), dtype=torch.int64, device=dets.device)
    nkeep = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    nbytes = L.lib().odw_nms_workspace(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dets.device)
    L.check(L.lib().odw_nms(L.ptr(dets), L.ptr(scores), n, float(threshold), _NMS_MODE[mode], L.ptr(keep),
                            L.ptr(nkeep), L.ptr(ws), nbytes, L.stream()), "nms")
    return keep[: int(nkeep.item())]


def nms(dets, scores, threshold):
    """wetectron `_C.nms` on a GPU tensor: +1 areas, `>` rule, ascending kept indices
    (csrc/nms.h:10-28 -> csrc/cuda/nms.cu:60,127-130)."""
    return _nms(dets, scores, threshold, "wt_gt")


def nms_cpu_rule(dets, scores, threshold):
    """Same, with the `>=` rule of csrc/cpu/nms_cpu.cpp:60."""
    return _nms(dets, scores, threshold, "wt_ge")


def nms_torchvision(boxes, scores, iou_threshold):
    """torchvision.ops.nms semantics (the reference's live path, structures/boxlist_ops.py:57)."""
    return _nms(boxes, scores, iou_threshold, "tv")


def box_iou(a, b):
    """boxlist_iou on raw (N,4)/(M,4) xyxy tensors (structures/boxlist_ops.py:127-160)."""
    L.need_gpu(a, b)
    a = a.contiguous().float()
    b = b.contiguous().float()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    if out.numel():
        L.check(L.lib().odw_box_iou(L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], L.ptr(out), L.stream()),
                "box_iou")
    return out


def pairwise_sim(E, padded=False):
    """E E^T (roi_heads/weak_head/loss.py:319), dense like torch.mm's result.

    The kernel is bound by the write of S and writes 128-byte row segments, which are whole cache lines only when the row
    pitch of S is a multiple of 16 floats.  ``padded=True`` returns the (P, P) view of a buffer whose rows are padded to a
    multiple of 32 floats -- the same values, ~15% sooner when P % 16 != 0 (P = 5000: 29.5 us against 33.9 us), for callers
    that only index S (``sim_mat[max_index]``, loss.py:320)."""
    L.need_gpu(E)
    E = E.contiguous().float()
    P, D = E.shape
    ld = -(-P // 32) * 32 if (padded and D == 128) else P
    buf = torch.empty((P, ld), dtype=torch.float32, device=E.device)
    if P and ld == P and D == 128 and P >= L.lib().odw_pairwise_sim_planes_min():
        # (only when ODW_PAIRWISE_PLANES_MIN forces it: the split + LDS-DMA pair, measured behind the one-launch panel kernel
        # at every size -- profiles/r06/pairwise_min_ab.txt)
        nbytes = L.lib().odw_pairwise_sim_workspace(P, D)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=E.device)
        L.check(L.lib().odw_pairwise_sim_ws(L.ptr(E), P, D, L.ptr(buf), L.ptr(ws), nbytes, L.stream()), "pairwise_sim")
    elif P:
        L.check(L.lib().odw_pairwise_sim_ld(L.ptr(E), P, D, L.ptr(buf), ld, L.stream()), "pairwise_sim")
    return buf if ld == P else buf[:, :P]


def supcon_v2(F, labels, weights, temperature, grad_scale=1.0, need_grad=True):
    """-> (loss scalar tensor, dF or None).  roi_heads/sim_head/sim_loss.py:49-80."""
    L.need_gpu(F, labels, weights)
    F = F.contiguous().float()
    labels = labels.contiguous().to(torch.int32)
    weights = weights.contiguous().float()
    N, D = F.shape
    loss = torch.empty((1,), dtype=torch.float32, device=F.device)
    dF = torch.empty_like(F) if need_grad else None
    nbytes = L.lib().odw_supcon_workspace(N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=F.device)
    L.check(L.lib().odw_supcon_v2(L.ptr(F), L.ptr(labels), L.ptr(weights), N, D, float(temperature),
                                  float(grad_scale), L.ptr(loss), L.ptr(dF), L.ptr(ws), nbytes, L.stream()),
            "supcon_v2")
    return loss[0], dF


def od_assign(boxes, gt_boxes, gt_classes, gt_scores, fg_thresh=0.5, weights=(10.0, 10.0, 5.0, 5.0), out=None):
    """-> (pseudo_labels int64 (P), loss_weights (P), regression_targets (P,4)).
    roi_heads/weak_head/pseudo_label_generator.py:171-190."""
    L.need_gpu(boxes, gt_boxes, gt_classes, gt_scores)
    boxes = boxes.contiguous().float()
    gt_boxes = gt_boxes.contiguous().float()
    gt_classes = gt_classes.contiguous().to(torch.int64)
    gt_scores = gt_scores.contiguous().float()
    P, G = boxes.shape[0], gt_boxes.shape[0]
    if out is not None:          # caller-provided contiguous slices (labels int64 (P), weights (P), targets (P,4))
        labels, lw, tg = out
    else:
        labels = torch.empty((P,), dtype=torch.int64, device=boxes.device)
        lw = torch.empty((P,), dtype=torch.float32, device=boxes.device)
        tg = torch.empty((P, 4), dtype=torch.float32, device=boxes.device)
    L.check(L.lib().odw_od_assign(L.ptr(boxes), P, L.ptr(gt_boxes), L.ptr(gt_classes), L.ptr(gt_scores), G,
                                  float(fg_thresh), *[float(w) for w in weights], L.ptr(labels), L.ptr(lw),
                                  L.ptr(tg), L.stream()), "od_assign")
    return labels, lw, tg


def image_preprocess(pixels, out_hw, out_plane, mean, std, to_bgr255=True, hflip=False, vflip=False, lighting=None):
    """Decoded uint8 image (H,W,3 RGB, on the GPU) -> its normalised, zero-padded fp32 slot (3,Hp,Wp) of the batch:
    Pillow-exact bilinear resize to out_hw, flips, /255, + lighting, BGR*255, (x-mean)/std in one pass
    (data/transforms/transforms.py:33-150 + structures/image_list.py:60-72)."""
    import ctypes
    import numpy as np
    L.need_gpu(pixels, out_plane)
    if pixels.dtype != torch.uint8 or pixels.dim() != 3 or pixels.shape[2] != 3 or not pixels.is_contiguous():
        raise ValueError("image_preprocess: pixels must be a contiguous uint8 (H,W,3) tensor")
    if out_plane.dtype != torch.float32 or out_plane.dim() != 3 or out_plane.shape[0] != 3 or not out_plane.is_contiguous():
        raise ValueError("image_preprocess: out_plane must be a contiguous fp32 (3,Hp,Wp) tensor")
    in_h, in_w = int(pixels.shape[0]), int(pixels.shape[1])
    oh, ow = int(out_hw[0]), int(out_hw[1])
    f3 = ctypes.c_float * 3
    mean_c = f3(*[float(np.float32(v)) for v in mean])
    std_c = f3(*[float(np.float32(v)) for v in std])
    light_c = f3(*[float(np.float32(v)) for v in lighting]) if lighting is not None else None
    nbytes = L.lib().odw_image_preprocess_workspace(in_h, in_w, oh, ow)
    ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=pixels.device)
    L.check(L.lib().odw_image_preprocess(
        L.ptr(pixels), in_h, in_w, oh, ow, int(bool(hflip)), int(bool(vflip)),
        ctypes.cast(light_c, ctypes.c_void_p) if light_c is not None else ctypes.c_void_p(0),
        ctypes.cast(mean_c, ctypes.c_void_p), ctypes.cast(std_c, ctypes.c_void_p), int(bool(to_bgr255)),
        L.ptr(out_plane), int(out_plane.shape[1]), int(out_plane.shape[2]), L.ptr(ws), int(nbytes), L.stream()),
        "image_preprocess")
    return out_plane
