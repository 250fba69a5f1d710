"""boxlist_iou / boxlist_nms_index / cat_boxlist on the gfx950 kernels
(wetectron/structures/boxlist_ops.py:38-61,127-160)."""
import torch

from .bounding_box import BoxList
from .. import _C


def boxlist_iou(boxlist1, boxlist2):
    """(N,M) IoU with the +1 pixel convention (TO_REMOVE = 1)."""
    if boxlist1.size != boxlist2.size:
        raise RuntimeError("boxlists should have same image size, got {}, {}".format(boxlist1, boxlist2))
    return _C.box_iou(boxlist1.convert("xyxy").bbox, boxlist2.convert("xyxy").bbox)


def boxlist_nms_index(boxlist, nms_thresh, max_proposals=-1, score_field="scores"):
    """-> (kept BoxList, keep indices in descending-score order); torchvision semantics
    (the reference calls torchvision.ops.nms here, boxlist_ops.py:57)."""
    if nms_thresh <= 0:
        return boxlist      # Q11: a bare BoxList, as the reference returns
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = _C.nms_torchvision(boxlist.bbox, boxlist.get_field(score_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep].convert(mode), keep


def remove_small_boxes(boxlist, min_size):
    """Keep boxes whose width AND height (+1 convention) are >= min_size (boxlist_ops.py:96-113)."""
    wh = boxlist.convert("xywh").bbox
    keep = torch.nonzero((wh[:, 2] >= min_size) & (wh[:, 3] >= min_size)).squeeze(1)
    return boxlist[keep]


def remove_small_area(boxlist, min_area):
    keep = torch.nonzero(boxlist.area() >= min_area).squeeze(1)
    return boxlist[keep], keep


def cat_boxlist(bboxes):
    size, mode = bboxes[0].size, bboxes[0].mode
    out = BoxList(torch.cat([b.bbox for b in bboxes], dim=0), size, mode)
    for f in bboxes[0].fields():
        out.add_field(f, torch.cat([b.get_field(f) for b in bboxes], dim=0))
    return out
