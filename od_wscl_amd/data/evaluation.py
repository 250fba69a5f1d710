"""PASCAL VOC detection metric (wetectron/data/datasets/evaluation/voc/voc_eval.py:12-287, itself the chainercv
port of the VOC devkit's evaluation): predictions are resized to the annotated image size, matched per class in
descending score against the ground truth at IoU >= 0.5 (integer-pixel convention: +1 on the far corner, `difficult`
objects neither reward nor punish, every ground-truth box is credited once), AP with the 11-point VOC07 rule.
Host-side numpy over CPU copies of the detections -- it runs once per evaluation, outside the hot path."""
import os
from collections import defaultdict

import numpy as np


def _iou_plus_one(a, b):
    """IoU with the +1 pixel convention (structures/boxlist_ops.py:127-160) of (n,4) x (m,4) arrays."""
    area_a = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    area_b = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt + 1, 0, None)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def calc_detection_voc_prec_rec(gt_boxlists, pred_boxlists, iou_thresh=0.5):
    n_pos, score, match = defaultdict(int), defaultdict(list), defaultdict(list)
    for gt, pred in zip(gt_boxlists, pred_boxlists):
        pred_bbox = pred.bbox.detach().cpu().numpy()
        pred_label = pred.get_field("labels").detach().cpu().numpy()
        pred_score = pred.get_field("scores").detach().cpu().numpy()
        gt_bbox = gt.bbox.detach().cpu().numpy()
        gt_label = gt.get_field("labels").detach().cpu().numpy()
        gt_difficult = gt.get_field("difficult").detach().cpu().numpy()
        for l in np.unique(np.concatenate((pred_label, gt_label)).astype(int)):
            sel = pred_label == l
            order = pred_score[sel].argsort()[::-1]
            pb, ps = pred_bbox[sel][order], pred_score[sel][order]
            gb, gd = gt_bbox[gt_label == l], gt_difficult[gt_label == l]
            n_pos[l] += np.logical_not(gd).sum()
            score[l].extend(ps)
            if len(pb) == 0:
                continue
            if len(gb) == 0:
                match[l].extend((0,) * pb.shape[0])
                continue
            pb, gb = pb.copy(), gb.copy()
            pb[:, 2:] += 1                       # "VOC evaluation follows integer typed bounding boxes"
            gb[:, 2:] += 1
            iou = _iou_plus_one(pb.astype(np.float32), gb.astype(np.float32))
            gt_index = iou.argmax(axis=1)
            gt_index[iou.max(axis=1) < iou_thresh] = -1
            taken = np.zeros(gb.shape[0], dtype=bool)
            for g in gt_index:
                if g < 0:
                    match[l].append(0)
                    continue
                match[l].append(-1 if gd[g] else (0 if taken[g] else 1))
                taken[g] = True
    n_fg_class = max(n_pos.keys()) + 1
    prec, rec = [None] * n_fg_class, [None] * n_fg_class
    for l in n_pos.keys():
        m = np.array(match[l], dtype=np.int8)[np.array(score[l]).argsort()[::-1]]
        tp, fp = np.cumsum(m == 1), np.cumsum(m == 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            prec[l] = tp / (fp + tp)
        if n_pos[l] > 0:
            rec[l] = tp / n_pos[l]
    return prec, rec


def calc_detection_voc_ap(prec, rec, use_07_metric=False):
    ap = np.empty(len(prec))
    for l in range(len(prec)):
        if prec[l] is None or rec[l] is None:
            ap[l] = np.nan
            continue
        if use_07_metric:
            ap[l] = 0
            for t in np.arange(0.0, 1.1, 0.1):
                p = 0 if np.sum(rec[l] >= t) == 0 else np.max(np.nan_to_num(prec[l])[rec[l] >= t])
                ap[l] += p / 11
        else:
            mpre = np.concatenate(([0], np.nan_to_num(prec[l]), [0]))
            mrec = np.concatenate(([0], rec[l], [1]))
            mpre = np.maximum.accumulate(mpre[::-1])[::-1]
            i = np.where(mrec[1:] != mrec[:-1])[0]
            ap[l] = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    return ap


def eval_detection_voc(pred_boxlists, gt_boxlists, iou_thresh=0.5, use_07_metric=False):
    assert len(gt_boxlists) == len(pred_boxlists), "Length of gt and pred lists need to be same."
    prec, rec = calc_detection_voc_prec_rec(gt_boxlists, pred_boxlists, iou_thresh)
    ap = calc_detection_voc_ap(prec, rec, use_07_metric=use_07_metric)
    return {"ap": ap, "map": np.nanmean(ap)}


def do_voc_evaluation(dataset, predictions, output_folder=None, logger=None):
    """voc_eval.py:12-43."""
    pred_boxlists, gt_boxlists = [], []
    for image_id, prediction in enumerate(predictions):
        info = dataset.get_img_info(image_id)
        pred_boxlists.append(prediction.resize((info["width"], info["height"])))
        gt_boxlists.append(dataset.get_groundtruth(image_id))
    result = eval_detection_voc(pred_boxlists, gt_boxlists, iou_thresh=0.5, use_07_metric=True)
    result_str = "mAP: {:.4f}\n".format(result["map"])
    for i, ap in enumerate(result["ap"]):
        if i == 0:
            continue
        result_str += "{:<16}: {:.4f}\n".format(dataset.map_class_id_to_class_name(i), ap)
    if logger is not None:
        logger.info(result_str)
    if output_folder:
        with open(os.path.join(output_folder, "result.txt"), "w") as fid:
            fid.write(result_str)
    return result


def evaluate(dataset, predictions, output_folder=None, task="det", logger=None, **_):
    """data/datasets/evaluation/__init__.py:6-28 for the VOC family."""
    if task != "det":
        raise NotImplementedError("only the detection mAP task is built (task=%r)" % task)
    return do_voc_evaluation(dataset, predictions, output_folder, logger)
