"""`nms` as exported by wetectron/layers/nms.py:6 (wetectron semantics), plus the
torchvision-semantics entry the reference's hot path actually calls
(structures/boxlist_ops.py:9,57)."""
from .. import _C


def nms(dets, scores, threshold):
    return _C.nms(dets.float(), scores.float(), threshold)


def nms_torchvision(boxes, scores, iou_threshold):
    return _C.nms_torchvision(boxes.float(), scores.float(), iou_threshold)
