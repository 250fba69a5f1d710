"""CPU restatement of the inference tail of the ROI head -- TEST INFRASTRUCTURE ONLY.

PostProcessor.forward + filter_results (wetectron/modeling/roi_heads/box_head/inference.py:41-90,216-258),
BoxCoder.decode (modeling/box_coder.py:52-95), BoxList.clip_to_image (structures/bounding_box.py:218-229),
boxlist_nms -> torchvision.ops.nms semantics (structures/boxlist_ops.py:13-36), reached from
ROIWeakRegHead.testing_forward "AVG" (roi_heads/weak_head/weak_head.py:131-134).
Pinned by tests/golden/infer_voc_2img.npz (the imported reference's own eval forward)."""
import math

import torch

from . import native
from . import hotpath_ref as H


def decode(rel_codes, boxes, weights=(10.0, 10.0, 5.0, 5.0), clip=math.log(1000.0 / 16)):
    """box_coder.py:52-95."""
    widths = boxes[:, 2] - boxes[:, 0] + 1
    heights = boxes[:, 3] - boxes[:, 1] + 1
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = rel_codes[:, 0::4] / wx
    dy = rel_codes[:, 1::4] / wy
    dw = torch.clamp(rel_codes[:, 2::4] / ww, max=clip)
    dh = torch.clamp(rel_codes[:, 3::4] / wh, max=clip)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    out = torch.zeros_like(rel_codes)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw - 1
    out[:, 3::4] = pcy + 0.5 * ph - 1
    return out


def filter_results(boxes, scores, size, score_thresh, nms_thr, max_det):
    """boxes (P, 4C) decoded + clipped, scores (P, C) -> (boxes (n,4), scores (n), labels (n)) class-major."""
    C = scores.shape[1]
    ob, os_, ol = [], [], []
    inds_all = scores > score_thresh
    for j in range(1, C):
        inds = inds_all[:, j].nonzero(as_tuple=False).squeeze(1)
        sj = scores[inds, j]
        bj = boxes[inds, j * 4:(j + 1) * 4]
        keep = torch.from_numpy(native.nms_tv(bj.contiguous().numpy(), sj.contiguous().numpy(), nms_thr))
        ob.append(bj[keep])
        os_.append(sj[keep])
        ol.append(torch.full((keep.numel(),), j, dtype=torch.int64))
    b, s, l = torch.cat(ob), torch.cat(os_), torch.cat(ol)
    n = s.numel()
    if n > max_det > 0:
        thresh, _ = torch.kthvalue(s, n - max_det + 1)
        keep = torch.nonzero(s >= thresh.item(), as_tuple=False).squeeze(1)
        b, s, l = b[keep], s[keep], l[keep]
    return b, s, l


def forward_eval(images, boxes_per_image, sizes_wh, sd, cfg, raw=False):
    """GeneralizedRCNN.forward in eval mode with precomputed proposals: backbone -> ROIPool -> fc6/fc7 (dropout is
    the identity) -> MISTPredictor eval branch (softmax-ed refinement scores, roi_weak_predictors.py:167-181) ->
    testing_forward "AVG" -> PostProcessor."""
    arch = cfg.get("arch", "vgg16")
    feat = H.backbone_forward(images, sd) if arch == "vgg16" else H.resnet_forward(images, sd, arch)
    rois = H.rois_with_batch_index(boxes_per_image)
    pooled = H._RoiPoolFn.apply(feat, rois, 7, 7, cfg.get("scale", 0.125))
    fe = "roi_heads.feature_extractor.classifier."
    a, b = ("1", "4") if (fe + "1.weight") in sd else ("0", "3")
    x = pooled.reshape(pooled.shape[0], -1)
    x = torch.relu(torch.nn.functional.linear(x, sd[fe + a + ".weight"], sd[fe + a + ".bias"]))
    x = torch.relu(torch.nn.functional.linear(x, sd[fe + b + ".weight"], sd[fe + b + ".bias"]))
    cls, det, refs, regs = H.predictor(x, sd)
    final_score = torch.mean(torch.stack([torch.softmax(r, dim=1) for r in refs]), dim=0)
    final_reg = torch.mean(torch.stack(regs), dim=0)
    all_boxes = torch.cat(boxes_per_image)
    dec = decode(final_reg, all_boxes)
    out, o = [], 0
    for bx, (w, h) in zip(boxes_per_image, sizes_wh):
        n = bx.shape[0]
        d = dec[o:o + n].reshape(-1, 4).clone()
        d[:, 0].clamp_(min=0, max=w - 1)
        d[:, 1].clamp_(min=0, max=h - 1)
        d[:, 2].clamp_(min=0, max=w - 1)
        d[:, 3].clamp_(min=0, max=h - 1)
        if raw:     # PostProcessor with bbox_aug_enabled (inference.py:85-88): decoded + clipped, not filtered
            out.append((d.reshape(n, -1), final_score[o:o + n]))
        else:
            out.append(filter_results(d.reshape(n, -1), final_score[o:o + n], (w, h), cfg.get("score_thresh", 0.0),
                                      cfg.get("nms_test", 0.4), cfg.get("max_det", 100)))
        o += n
    return out


def tta(pixels_list, raw_boxes_list, sd, cfg, aug):
    """im_detect_bbox_aug (wetectron/engine/bbox_aug.py:11-77) for a BATCH of images (the passes run on the whole
    batch, so the zero padding to the batch's common size is part of the result): the identity pass, its flip, every
    scale (+ flip); boxes un-flipped (W - x - 1) and resized to the first pass's frame; "AVG" merge; filter_results.
    pixels uint8 (H,W,3) per image; raw_boxes (n,4) fp32 in the original frame; aug = dict(min_test, max_test, h_flip,
    scales, max_size, scale_h_flip, mean, std, to_bgr255, size_divisible)."""
    import numpy as np
    from . import data_ref as D
    passes = [(aug["min_test"], aug["max_test"], False)]
    if aug["h_flip"]:
        passes.append((aug["min_test"], aug["max_test"], True))
    for s in aug["scales"]:
        passes.append((s, aug["max_size"], False))
        if aug["scale_h_flip"]:
            passes.append((s, aug["max_size"], True))
    n_img = len(pixels_list)
    merged_b, merged_s, first = [[] for _ in range(n_img)], [[] for _ in range(n_img)], [None] * n_img
    for size, max_size, flip in passes:
        imgs, boxes, sizes = [], [], []
        for pixels, raw_boxes in zip(pixels_list, raw_boxes_list):
            h0, w0 = pixels.shape[:2]
            oh, ow = D.get_size((w0, h0), size, max_size)
            imgs.append(D.pixel_chain(pixels, (oh, ow), flip, False, None, aug["mean"], aug["std"], aug["to_bgr255"]))
            b = D.boxes_resize(raw_boxes, (w0, h0), (ow, oh))
            if flip:
                b = D.boxes_transpose(b, (ow, oh), 0)
            boxes.append(torch.from_numpy(np.ascontiguousarray(b)))
            sizes.append((ow, oh))
        batch, _ = D.to_image_list(imgs, aug["size_divisible"])
        raw = forward_eval(torch.from_numpy(batch), boxes, sizes, sd, cfg, raw=True)
        for i, (dec, sc) in enumerate(raw):
            dec = dec.reshape(-1, 4).numpy()
            if flip:
                dec = D.boxes_transpose(dec, sizes[i], 0)
            if first[i] is None:
                first[i] = sizes[i]
            else:
                dec = D.boxes_resize(dec, sizes[i], first[i])
            merged_b[i].append(torch.from_numpy(np.ascontiguousarray(dec)))
            merged_s[i].append(sc.reshape(-1))
            C = sc.shape[1]
    out = []
    for i in range(n_img):
        if aug.get("heur", "AVG") == "UNION":       # bbox_aug.py:57-59: every pass's (P*C) boxes side by side
            bbox = torch.cat(merged_b[i])
            scores = torch.cat(merged_s[i])
        else:
            bbox = torch.mean(torch.stack(merged_b[i]), dim=0)
            scores = torch.mean(torch.stack(merged_s[i]), dim=0)
        out.append(filter_results(bbox.reshape(-1, C * 4), scores.reshape(-1, C), first[i], cfg.get("score_thresh", 0.0),
                                  cfg.get("nms_test", 0.4), cfg.get("max_det", 100)))
    return out
