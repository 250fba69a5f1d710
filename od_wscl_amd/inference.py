"""Evaluation loop (wetectron/engine/inference.py:19-139): every rank runs its shard of the test set (single scale, or
test-time augmentation when TEST.BBOX_AUG.ENABLED), the per-image detections are gathered on rank 0
(`all_gather_object` over the process group -- no tensor collective on this path), saved as predictions.pth and
scored with the VOC metric."""
import logging
import os
import time

import torch
import torch.distributed as dist

from .bbox_aug import im_detect_bbox_aug
from .data.evaluation import evaluate


def compute_on_dataset(model, data_loader, device, cfg, timer=None):
    model.eval()
    results = {}
    cpu = torch.device("cpu")
    for batch in data_loader:
        images, targets, rois, image_ids = batch
        with torch.no_grad():
            t0 = time.time()
            if cfg.TEST.BBOX_AUG.ENABLED:
                output = im_detect_bbox_aug(model, images, device, rois, cfg)
            else:
                rois = [r.to(device) if r is not None else None for r in rois]
                output = model(images.to(device), rois=rois)
            if timer is not None:
                torch.cuda.synchronize()
                timer.append(time.time() - t0)
            output = [o.to(cpu) for o in output]
        results.update({img_id: result for img_id, result in zip(image_ids, output)})
    return results


def _accumulate_predictions_from_multiple_gpus(predictions_per_gpu):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, predictions_per_gpu)
        if dist.get_rank() != 0:
            return None
    else:
        gathered = [predictions_per_gpu]
    predictions = {}
    for p in gathered:
        predictions.update(p)
    image_ids = list(sorted(predictions.keys()))
    if len(image_ids) != image_ids[-1] + 1:
        logging.getLogger("od_wscl_amd.inference").warning(
            "Number of images that were gathered from multiple processes is not a contiguous set. "
            "Some images might be missing from the evaluation")
    return [predictions[i] for i in image_ids]


def inference(model, data_loader, dataset_name, cfg, device="cuda", output_folder=None, task="det"):
    device = torch.device(device)
    logger = logging.getLogger("od_wscl_amd.inference")
    dataset = data_loader.dataset
    logger.info("Start evaluation on {} dataset({} images).".format(dataset_name, len(dataset)))
    saved = os.path.join(output_folder, "predictions.pth") if output_folder else None
    if saved and os.path.exists(saved):
        predictions = torch.load(saved, weights_only=False)
    else:
        timer = []
        t0 = time.time()
        predictions = compute_on_dataset(model, data_loader, device, cfg, timer)
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        logger.info("Total run time: {:.1f} s ({:.4f} s / img per device, on {} devices); model time {:.4f} s / img".format(
            time.time() - t0, (time.time() - t0) * world / max(len(dataset), 1), world,
            sum(timer) * world / max(len(dataset), 1)))
        predictions = _accumulate_predictions_from_multiple_gpus(predictions)
        if predictions is None:
            return None
        if saved:
            torch.save(predictions, saved)
    return evaluate(dataset=dataset, predictions=predictions, output_folder=output_folder, task=task, logger=logger)
