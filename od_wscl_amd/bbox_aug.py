"""Test-time augmentation (wetectron/engine/bbox_aug.py:11-137): the detector runs on the original scale, its
horizontal flip, and every TEST.BBOX_AUG.SCALES scale (+ flips); the un-filtered (P*C) boxlists are brought back to
the first pass's frame (un-flip, resize), merged ("AVG": mean of boxes and scores; "UNION": concatenation) and only
then filtered (score threshold, per-class NMS, best 100).

MI355X shape of it: each image's decoded pixels are uploaded ONCE and every pass re-runs the fused preprocessing
kernel (csrc/preprocess.hip) from them -- 14 passes of the shipped VOC config cost one 0.5 MB copy instead of 14
host-side PIL resizes and 14 fp32 uploads; decoding for all classes is one launch per pass (odw_detect_decode), the
final filter one launch per image (odw_detect_filter)."""
import torch

from .data import transforms as T
from .modeling.roi_heads.box_head.inference import make_roi_box_post_processor
from .structures.bounding_box import BoxList, FLIP_LEFT_RIGHT
from .structures.image_list import to_image_list


def im_detect_bbox_aug(model, images, device, rois=None, cfg=None):
    """images: the raw PIL images / uint8 arrays of a BBoxAugCollator batch, rois: their proposals (BoxList, original
    frame).  `model` has to be in eval mode and built with TEST.BBOX_AUG.ENABLED (its post-processor then returns the
    decoded, un-filtered boxlists)."""
    if cfg is None:
        from .config import cfg
    base = [T.defer(im) for im in images]          # decoded pixels, shared by every pass (device copy cached)
    boxlists_ts = [[] for _ in base]

    def add_preds_t(boxlists_t):
        for i, boxlist_t in enumerate(boxlists_t):
            if len(boxlists_ts[i]) == 0:
                boxlists_ts[i].append(boxlist_t)       # identity transform: already in the target frame
            else:
                boxlists_ts[i].append(boxlist_t.resize(boxlists_ts[i][0].size))

    add_preds_t(im_detect_bbox(model, base, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, device, cfg, rois=rois))
    if cfg.TEST.BBOX_AUG.H_FLIP:
        add_preds_t(im_detect_bbox_hflip(model, base, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, device, cfg,
                                         rois=rois))
    for scale in cfg.TEST.BBOX_AUG.SCALES:
        max_size = cfg.TEST.BBOX_AUG.MAX_SIZE
        add_preds_t(im_detect_bbox_scale(model, base, scale, max_size, device, cfg, rois=rois))
        if cfg.TEST.BBOX_AUG.SCALE_H_FLIP:
            add_preds_t(im_detect_bbox_scale(model, base, scale, max_size, device, cfg, hflip=True, rois=rois))

    post_processor = make_roi_box_post_processor(cfg)
    results = []
    for boxlist_ts in boxlists_ts:
        if cfg.TEST.BBOX_AUG.HEUR == "UNION":
            bbox = torch.cat([b.bbox for b in boxlist_ts])
            scores = torch.cat([b.get_field("scores") for b in boxlist_ts])
        elif cfg.TEST.BBOX_AUG.HEUR == "AVG":
            bbox = torch.mean(torch.stack([b.bbox for b in boxlist_ts]), dim=0)
            scores = torch.mean(torch.stack([b.get_field("scores") for b in boxlist_ts]), dim=0)
        else:
            raise ValueError("please use proper BBOX_AUG.HEUR ")
        boxlist = BoxList(bbox, boxlist_ts[0].size, boxlist_ts[0].mode)
        boxlist.add_field("scores", scores)
        results.append(post_processor.filter_results(boxlist, cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES))
    return results


def _run(model, base, target_scale, target_max_size, device, cfg, rois, hflip):
    chain = [T.Resize(target_scale, target_max_size)]
    if hflip:
        chain.append(T.RandomHorizontalFlip(1.0))
    chain += [T.ToTensor(), T.Normalize(mean=cfg.INPUT.PIXEL_MEAN, std=cfg.INPUT.PIXEL_STD, to_bgr255=cfg.INPUT.TO_BGR255)]
    transform = T.Compose(chain)
    t_images, t_rois = [], []
    for image, roi in zip(base, rois):
        t_img, _, t_roi = transform(image.fork(), rois=roi)
        t_images.append(t_img)
        t_rois.append(t_roi)
    t_images = to_image_list(t_images, cfg.DATALOADER.SIZE_DIVISIBILITY)
    t_rois = [r.to(device) if r is not None else None for r in t_rois]
    return model(t_images.to(device), rois=t_rois)


def im_detect_bbox(model, images, target_scale, target_max_size, device, cfg, rois=None):
    """bbox_aug.py:81-103."""
    return _run(model, images, target_scale, target_max_size, device, cfg, rois, hflip=False)


def im_detect_bbox_hflip(model, images, target_scale, target_max_size, device, cfg, rois=None):
    """bbox_aug.py:106-131: detect on the mirrored image, mirror the detections back."""
    boxlists = _run(model, images, target_scale, target_max_size, device, cfg, rois, hflip=True)
    return [boxlist.transpose(FLIP_LEFT_RIGHT) for boxlist in boxlists]


def im_detect_bbox_scale(model, images, target_scale, target_max_size, device, cfg, hflip=False, rois=None):
    """bbox_aug.py:134-144."""
    if hflip:
        return im_detect_bbox_hflip(model, images, target_scale, target_max_size, device, cfg, rois=rois)
    return im_detect_bbox(model, images, target_scale, target_max_size, device, cfg, rois=rois)
