"""PostProcessor -- detections from class scores, box regression and proposals
(wetectron/modeling/roi_heads/box_head/inference.py:12-90,216-282; the weak variant without regression:
roi_heads/weak_head/inference.py:10-134).

decode -> clip -> per-class (score > thresh, NMS) -> labels -> keep the best `detections_per_img` over all classes.
On the GPU the first four stages are ONE kernel launch (csrc/detect.hip, one workgroup per (image, class)) and
one small blocking read of the per-class counts; the reference runs a Python loop over the classes with a
nonzero() synchronisation and an NMS launch per class."""
import torch
import torch.nn.functional as F
from torch import nn

from .... import _lib as L
from ....structures import BoxList
from ...box_coder import BoxCoder


class PostProcessor(nn.Module):
    def __init__(self, score_thresh=0.05, nms=0.5, detections_per_img=100, box_coder=None,
                 cls_agnostic_bbox_reg=False, bbox_aug_enabled=False, regression=True):
        super().__init__()
        self.score_thresh, self.nms, self.detections_per_img = score_thresh, nms, detections_per_img
        self.box_coder = box_coder if box_coder is not None else BoxCoder(weights=(10.0, 10.0, 5.0, 5.0))
        self.cls_agnostic_bbox_reg, self.bbox_aug_enabled, self.regression = cls_agnostic_bbox_reg, bbox_aug_enabled, regression

    def forward(self, x, boxes, softmax_on=True):
        """x = (class scores (sumP, C), box regression (sumP, 4C | 4)) -- or the scores alone for the weak variant;
        boxes: list[BoxList] proposals.  Returns one BoxList per image with fields `scores`, `labels`."""
        if self.regression:
            class_logits, box_regression = x
        else:
            class_logits, box_regression, softmax_on = x, None, False
        class_prob = F.softmax(class_logits, -1) if softmax_on else class_logits
        L.need_gpu(class_prob, boxes[0].bbox)
        dev = class_prob.device
        sizes = [len(b) for b in boxes]
        n_img, C, max_p = len(boxes), class_prob.shape[1], max(sizes)
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        concat = torch.cat([b.bbox for b in boxes], dim=0).float().contiguous()
        prob = class_prob.float().contiguous()
        reg = None
        if box_regression is not None:
            reg = box_regression.reshape(offs[-1], -1).float()
            if self.cls_agnostic_bbox_reg:
                reg = reg[:, -4:]
            reg = reg.contiguous()
        img_off = torch.tensor(offs, dtype=torch.int32, device=dev)
        img_wh = torch.tensor([[float(b.size[0]), float(b.size[1])] for b in boxes], dtype=torch.float32, device=dev)
        w = self.box_coder.weights
        if self.bbox_aug_enabled:
            # inference.py:78-88: decoded + clipped (P*C) boxlists, filtered later by the test-time augmentation
            decoded = torch.empty((offs[-1], C, 4), dtype=torch.float32, device=dev)
            L.check(L.lib().odw_detect_decode(L.ptr(reg), reg.shape[1] if reg is not None else 0,
                                              1 if self.cls_agnostic_bbox_reg else 0, L.ptr(concat), L.ptr(img_off),
                                              L.ptr(img_wh), n_img, offs[-1], C, float(w[0]), float(w[1]), float(w[2]),
                                              float(w[3]), float(self.box_coder.bbox_xform_clip), L.ptr(decoded),
                                              L.stream()), "detect_decode")
            results = []
            for i, b in enumerate(boxes):
                r = BoxList(decoded[offs[i]:offs[i + 1]].reshape(-1, 4), b.size, mode="xyxy")
                r.add_field("scores", prob[offs[i]:offs[i + 1]].reshape(-1))
                results.append(r)
            return results
        if self.nms <= 0:
            raise ValueError("MODEL.ROI_HEADS.NMS must be > 0")
        if max_p > self.FUSED_MAX_P:
            # more proposals than one workgroup of the fused kernel sorts in LDS (the reference has no cap at test
            # time: top_k is -1 for COCO): decode on the device, then the reference's own per-class loop on odw_nms
            decoded = torch.empty((offs[-1], C, 4), dtype=torch.float32, device=dev)
            L.check(L.lib().odw_detect_decode(L.ptr(reg), reg.shape[1] if reg is not None else 0,
                                              1 if self.cls_agnostic_bbox_reg else 0, L.ptr(concat), L.ptr(img_off),
                                              L.ptr(img_wh), n_img, offs[-1], C, float(w[0]), float(w[1]), float(w[2]),
                                              float(w[3]), float(self.box_coder.bbox_xform_clip), L.ptr(decoded),
                                              L.stream()), "detect_decode")
            return [self._per_class(prob[offs[i]:offs[i + 1]], decoded[offs[i]:offs[i + 1]], b.size, C)
                    for i, b in enumerate(boxes)]
        out = self._outputs(n_img, C, max_p, dev)
        L.check(L.lib().odw_detect_postprocess(L.ptr(prob), C, L.ptr(reg), reg.shape[1] if reg is not None else 0,
                                               1 if self.cls_agnostic_bbox_reg else 0, L.ptr(concat), L.ptr(img_off),
                                               L.ptr(img_wh), n_img, max_p, float(w[0]), float(w[1]), float(w[2]), float(w[3]),
                                               float(self.box_coder.bbox_xform_clip), float(self.score_thresh), float(self.nms),
                                               max_p, L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), L.ptr(out[3]),
                                               L.stream()), "detect_postprocess")
        return self._collect(out, [b.size for b in boxes], C)

    FUSED_MAX_P = 4096          # boxes per (image, class) the fused kernels hold in LDS (csrc/detect.hip)

    def _per_class(self, prob, boxes_pc, size, C):
        """inference.py:216-258 class by class for box counts beyond the fused kernel: score > thresh, odw_nms with
        torchvision semantics, labels; then the same best-`detections_per_img` cut.  One host synchronisation per
        class, like the reference."""
        from .... import _C
        dev = prob.device
        bx, sc, lab, idx = [], [], [], []
        for j in range(1, C):
            inds = torch.nonzero(prob[:, j] > self.score_thresh, as_tuple=False).squeeze(1)
            if inds.numel() == 0:
                continue
            if inds.numel() > 8192:
                raise NotImplementedError("PostProcessor: %d boxes of one class pass the score threshold; odw_nms sorts "
                                          "at most 8192 (ODW_NMS_MAX_N)" % inds.numel())
            b_j, s_j = boxes_pc[inds, j].contiguous(), prob[inds, j].contiguous()
            keep = _C.nms_torchvision(b_j, s_j, self.nms)
            bx.append(b_j[keep]); sc.append(s_j[keep]); idx.append(inds[keep].int())
            lab.append(torch.full((keep.numel(),), j, dtype=torch.int64, device=dev))
        return self._finish(bx, sc, lab, idx, size, dev)

    @staticmethod
    def _outputs(n_img, C, max_p, dev):
        return (torch.empty((n_img, C - 1, max_p, 4), dtype=torch.float32, device=dev),
                torch.empty((n_img, C - 1, max_p), dtype=torch.float32, device=dev),
                torch.empty((n_img, C - 1, max_p), dtype=torch.int32, device=dev),
                torch.zeros((n_img, C - 1), dtype=torch.int32, device=dev))

    def filter_results(self, boxlist, num_classes):
        """inference.py:216-258 on one (P*C)-box list with (P*C) scores -- what the test-time augmentation hands over
        after merging its passes: per class score > thresh, NMS, labels, best `detections_per_img` overall."""
        boxes_pc = boxlist.bbox.reshape(-1, num_classes, 4).float().contiguous()
        prob = boxlist.get_field("scores").reshape(-1, num_classes).float().contiguous()
        L.need_gpu(boxes_pc, prob)
        if self.nms <= 0:
            raise ValueError("MODEL.ROI_HEADS.NMS must be > 0")
        P = boxes_pc.shape[0]
        if P > self.FUSED_MAX_P:        # e.g. the UNION heuristic of TEST.BBOX_AUG concatenates every pass
            return self._per_class(prob, boxes_pc, boxlist.size, num_classes)
        dev = prob.device
        img_off = torch.tensor([0, P], dtype=torch.int32, device=dev)
        out = self._outputs(1, num_classes, P, dev)
        L.check(L.lib().odw_detect_filter(L.ptr(prob), num_classes, L.ptr(boxes_pc), L.ptr(img_off), 1, P,
                                          float(self.score_thresh), float(self.nms), P, L.ptr(out[0]), L.ptr(out[1]),
                                          L.ptr(out[2]), L.ptr(out[3]), L.stream()), "detect_filter")
        return self._collect(out, [boxlist.size], num_classes)[0]

    def _collect(self, out, image_sizes, C):
        out_boxes, out_scores, out_index, out_count = out
        dev = out_boxes.device
        counts = out_count.cpu().numpy()                               # the one blocking read
        results = []
        for i, size in enumerate(image_sizes):
            bx, sc, lab, idx = [], [], [], []
            for j in range(1, C):
                k = int(counts[i, j - 1])
                if k:
                    bx.append(out_boxes[i, j - 1, :k])
                    sc.append(out_scores[i, j - 1, :k])
                    idx.append(out_index[i, j - 1, :k])
                    lab.append(torch.full((k,), j, dtype=torch.int64, device=dev))
            results.append(self._finish(bx, sc, lab, idx, size, dev))
        return results

    def _finish(self, bx, sc, lab, idx, size, dev):
        """Per-class survivors -> one BoxList, cut to the best `detections_per_img` (inference.py:246-255: ties kept)."""
        if bx:
            bx, sc, lab, idx = torch.cat(bx), torch.cat(sc), torch.cat(lab), torch.cat(idx)
        else:
            bx, sc = torch.zeros((0, 4), device=dev), torch.zeros((0,), device=dev)
            lab, idx = torch.zeros((0,), dtype=torch.int64, device=dev), torch.zeros((0,), dtype=torch.int32, device=dev)
        n = int(sc.numel())
        if n > self.detections_per_img > 0:
            thresh, _ = torch.kthvalue(sc.cpu(), n - self.detections_per_img + 1)
            keep = torch.nonzero(sc >= thresh.item(), as_tuple=False).squeeze(1)
            bx, sc, lab, idx = bx[keep], sc[keep], lab[keep], idx[keep]
        r = BoxList(bx, size, mode="xyxy")
        r.add_field("scores", sc)
        r.add_field("labels", lab)
        r.add_field("proposal_index", idx.long())
        return r


def make_roi_box_post_processor(cfg, regression=True):
    """box_head/inference.py:268-282 (regression) and weak_head/inference.py:136-146 (scores only)."""
    return PostProcessor(cfg.MODEL.ROI_HEADS.SCORE_THRESH, cfg.MODEL.ROI_HEADS.NMS, cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG,
                         BoxCoder(weights=cfg.MODEL.ROI_HEADS.BBOX_REG_WEIGHTS), cfg.MODEL.CLS_AGNOSTIC_BBOX_REG,
                         cfg.TEST.BBOX_AUG.ENABLED, regression=regression)
