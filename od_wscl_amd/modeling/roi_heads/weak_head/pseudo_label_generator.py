"""od_layer / oicr_layer (wetectron/modeling/roi_heads/weak_head/pseudo_label_generator.py:83-197).

Pseudo ground truth = the boxes object discovery kept for each positive class (or the top-1
proposal), then a fused device kernel does IoU -> first-argmax -> labels / weights / targets.
The reference pulls the (P,G) IoU matrix to the host for numpy max/argmax (:176-177); nothing
leaves the device here."""
import torch

from .... import _C


class od_layer(object):
    def __init__(self, cfg):
        self.weights = tuple(cfg.MODEL.ROI_HEADS.BBOX_REG_WEIGHTS)
        self.fg_thresh = cfg.MODEL.ROI_HEADS.FG_IOU_THRESHOLD

    @torch.no_grad()
    def __call__(self, proposals, source_score, labels, device, pgt_instance, return_targets=False):
        prob = source_score[:, 1:].clone()
        boxes = proposals.bbox
        gt_b, gt_c, gt_s = [], [], []
        for c in labels[1:].eq(1).nonzero(as_tuple=False)[:, 0].tolist():
            col = prob[:, c]
            top = torch.argmax(col)
            picked = pgt_instance[c] if pgt_instance is not None else None
            if picked is None or picked.numel() == 0:
                picked = top.view(1)
            gt_b.append(boxes[picked])
            gt_c.append(torch.full((picked.numel(),), c + 1, dtype=torch.int64, device=boxes.device))
            gt_s.append(col[picked].clone())
            prob[top].fill_(0)          # Q5: the whole row of the top proposal is zeroed
        n = source_score.shape[0]
        if not gt_b:
            out = (torch.zeros(n, dtype=torch.int64, device=boxes.device),
                   torch.zeros(n, dtype=torch.float32, device=boxes.device))
            return out + (None,) if return_targets else out
        pseudo, weights, targets = _C.od_assign(boxes, torch.cat(gt_b), torch.cat(gt_c), torch.cat(gt_s),
                                                self.fg_thresh, self.weights)
        if return_targets:
            return pseudo, weights, targets
        return pseudo, weights


class oicr_layer(od_layer):
    """OICR (Tang et al. 2017): od_layer with the top-1 proposal as the only pseudo-GT per class
    (pseudo_label_generator.py:83-133)."""

    @torch.no_grad()
    def __call__(self, proposals, source_score, labels, device, return_targets=False):
        return od_layer.__call__(self, proposals, source_score, labels, device, None, return_targets)
