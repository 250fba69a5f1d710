"""make_data_loader (wetectron/data/build.py:19-232): datasets from a catalog, rank-sharded sampling, aspect-ratio /
class-pair batching, iteration-based resampling, collation.  The DataLoader workers decode images and run the
GEOMETRY of the transforms; what they ship to the trainer is uint8 pixels + a pixel plan per image
(data/transforms.DeferredImage) and the GPU does the pixel work at `.to(device)` (csrc/preprocess.hip)."""
import bisect
import copy
import os

import numpy as np
import torch
import torch.utils.data

from . import datasets as D
from . import samplers
from .collate_batch import BatchCollator, BBoxAugCollator
from .transforms import build_transforms


class DatasetCatalog(object):
    """The VOC and COCO entries of wetectron/config/paths_catalog.py:10-167."""
    DATA_DIR = "datasets"
    DATASETS = {
        "voc_2007_train": {"data_dir": "voc/VOC2007", "split": "train"},
        "voc_2007_val": {"data_dir": "voc/VOC2007", "split": "val"},
        "voc_2007_trainval": {"data_dir": "voc/VOC2007", "split": "trainval"},
        "voc_2007_test": {"data_dir": "voc/VOC2007", "split": "test"},
        "voc_2012_train": {"data_dir": "voc/VOC2012", "split": "train"},
        "voc_2012_val": {"data_dir": "voc/VOC2012", "split": "val"},
        "voc_2012_trainval": {"data_dir": "voc/VOC2012", "split": "trainval"},
        "voc_2012_test": {"data_dir": "voc/VOC2012", "split": "test"},
        "coco_2017_train": {"img_dir": "coco/train2017", "ann_file": "coco/annotations/instances_train2017.json"},
        "coco_2017_val": {"img_dir": "coco/val2017", "ann_file": "coco/annotations/instances_val2017.json"},
        "coco_2014_train": {"img_dir": "coco/train2014", "ann_file": "coco/annotations/instances_train2014.json"},
        "coco_2014_val": {"img_dir": "coco/val2014", "ann_file": "coco/annotations/instances_val2014.json"},
        "coco_2014_minival": {"img_dir": "coco/val2014", "ann_file": "coco/annotations/instances_minival2014.json"},
        "coco_2014_valminusminival": {"img_dir": "coco/val2014",
                                      "ann_file": "coco/annotations/instances_valminusminival2014.json"},
    }

    @classmethod
    def get(cls, name):
        if "coco" in name and name in cls.DATASETS:
            attrs = cls.DATASETS[name]
            return dict(factory="COCODataset", args=dict(root=os.path.join(cls.DATA_DIR, attrs["img_dir"]),
                                                         ann_file=os.path.join(cls.DATA_DIR, attrs["ann_file"])))
        if "voc" in name and name in cls.DATASETS:
            attrs = cls.DATASETS[name]
            return dict(factory="PascalVOCDataset",
                        args=dict(data_dir=os.path.join(cls.DATA_DIR, attrs["data_dir"]), split=attrs["split"]))
        raise RuntimeError("Dataset not available: {}".format(name))


def build_dataset(dataset_list, transforms, dataset_catalog, is_train=True, proposal_files=None, min_size=None):
    """build.py:19-79."""
    if not isinstance(dataset_list, (list, tuple)):
        raise RuntimeError("dataset_list should be a list of strings, got {}".format(dataset_list))
    if proposal_files is not None and len(proposal_files) == 0:
        proposal_files = (None,) * len(dataset_list)
    datasets, data_args = [], []
    for index, dataset_name in enumerate(dataset_list):
        data = dataset_catalog.get(dataset_name)
        factory = getattr(D, data["factory"])
        args = dict(data["args"])
        if data["factory"] == "COCODataset":
            args["remove_images_without_annotations"] = is_train and "unlabeled" not in dataset_name
        if data["factory"] == "PascalVOCDataset":
            args["use_difficult"] = not is_train
        args["transforms"] = transforms
        args["min_size"] = min_size
        if proposal_files is not None:
            args["proposal_file"] = proposal_files[index]
        datasets.append(factory(**args))
        data_args.append(args)
    if not is_train:
        return datasets, data_args
    dataset = datasets[0] if len(datasets) == 1 else D.ConcatDataset(datasets)
    return [dataset], [data_args]


def make_data_sampler(dataset, shuffle, distributed):
    if distributed:
        return samplers.DistributedSampler(dataset, shuffle=shuffle)
    if shuffle:
        return torch.utils.data.sampler.RandomSampler(dataset)
    return torch.utils.data.sampler.SequentialSampler(dataset)


def _quantize(x, bins):
    bins = sorted(copy.copy(bins))
    return [bisect.bisect_right(bins, y) for y in x]


def _compute_aspect_ratios(dataset):
    out = []
    for i in range(len(dataset)):
        info = dataset.get_img_info(i)
        out.append(float(info["height"]) / float(info["width"]))
    return out


def make_batch_data_sampler(dataset, sampler, aspect_grouping, images_per_batch, batch_size=None, args=None,
                            class_batch=False, num_iters=None, start_iter=0):
    """build.py:111-141."""
    if aspect_grouping:
        if not isinstance(aspect_grouping, (list, tuple)):
            aspect_grouping = [aspect_grouping]
        group_ids = _quantize(_compute_aspect_ratios(dataset), aspect_grouping)
        batch_sampler = samplers.GroupedBatchSampler(sampler, group_ids, images_per_batch, batch_size, dataset,
                                                     class_batch, data_args=args, drop_uneven=False)
    else:
        batch_sampler = torch.utils.data.sampler.BatchSampler(sampler, images_per_batch, drop_last=False)
    if num_iters is not None:
        batch_sampler = samplers.IterationBasedBatchSampler(batch_sampler, num_iters, start_iter)
    return batch_sampler


def worker_init_reset_seed(worker_id):
    import random
    seed = np.random.randint(2 ** 31) + worker_id
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)


def make_data_loader(cfg, is_train=True, is_distributed=False, start_iter=0, dataset_catalog=None, num_gpus=None):
    """build.py:143-229.  Returns one loader in training, a list (one per test set) otherwise."""
    if num_gpus is None:
        num_gpus = torch.distributed.get_world_size() if torch.distributed.is_available() and \
            torch.distributed.is_initialized() else 1
    if is_train:
        images_per_batch = cfg.SOLVER.IMS_PER_BATCH
        shuffle, num_iters = True, cfg.SOLVER.MAX_ITER
    else:
        images_per_batch = cfg.TEST.IMS_PER_BATCH
        shuffle, num_iters, start_iter = (False if not is_distributed else True), None, 0
    assert images_per_batch % num_gpus == 0, \
        "IMS_PER_BATCH ({}) must be divisible by the number of GPUs ({}) used.".format(images_per_batch, num_gpus)
    images_per_gpu = images_per_batch // num_gpus
    aspect_grouping = [1] if cfg.DATALOADER.ASPECT_RATIO_GROUPING else []
    catalog = dataset_catalog if dataset_catalog is not None else DatasetCatalog
    dataset_list = cfg.DATASETS.TRAIN if is_train else cfg.DATASETS.TEST
    proposal_files = cfg.PROPOSAL_FILES.TRAIN if is_train else cfg.PROPOSAL_FILES.TEST
    # with test-time augmentation the raw images travel and im_detect_bbox_aug runs the transforms (:193)
    transforms = None if not is_train and cfg.TEST.BBOX_AUG.ENABLED else build_transforms(cfg, is_train)
    datasets, data_args = build_dataset(dataset_list, transforms, catalog, is_train, proposal_files, cfg.min_size)
    class_batch = cfg.SOLVER.CLASS_BATCH if is_train else False
    data_loaders = []
    for dataset in datasets:
        sampler = make_data_sampler(dataset, shuffle, is_distributed)
        batch_sampler = make_batch_data_sampler(dataset, sampler, aspect_grouping, images_per_gpu, images_per_batch,
                                                data_args, class_batch, num_iters, start_iter)
        collator = BBoxAugCollator() if not is_train and cfg.TEST.BBOX_AUG.ENABLED else \
            BatchCollator(cfg.DATALOADER.SIZE_DIVISIBILITY)
        data_loaders.append(torch.utils.data.DataLoader(dataset, num_workers=cfg.DATALOADER.NUM_WORKERS,
                                                        batch_sampler=batch_sampler, collate_fn=collator,
                                                        worker_init_fn=worker_init_reset_seed))
    if is_train:
        assert len(data_loaders) == 1
        return data_loaders[0]
    return data_loaders
