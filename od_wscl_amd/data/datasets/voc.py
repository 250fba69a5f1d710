"""PascalVOCDataset (wetectron/data/datasets/voc.py:13-203): VOC devkit layout (JPEGImages / Annotations /
ImageSets/Main), image-level labels from the XML objects, proposals from a ProposalFile.  A sample is
(image, target, rois, index); with transforms the image is a `DeferredImage` (decoded pixels + pixel plan) that the
GPU turns into the normalised tensor at collation time."""
import os
import xml.etree.ElementTree as ET

import torch
import torch.utils.data
from PIL import Image

from ...structures.bounding_box import BoxList
from .proposals import ProposalFile, prepare_proposals


class PascalVOCDataset(torch.utils.data.Dataset):
    CLASSES = ("__background__ ", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair",
               "cow", "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train",
               "tvmonitor")

    def __init__(self, data_dir, split, use_difficult=False, transforms=None, proposal_file=None, min_size=None):
        self.root = data_dir
        self.image_set = split
        self.keep_difficult = use_difficult
        self.transforms = transforms
        self._annopath = os.path.join(self.root, "Annotations", "%s.xml")
        self._imgpath = os.path.join(self.root, "JPEGImages", "%s.jpg")
        self._imgsetpath = os.path.join(self.root, "ImageSets", "Main", "%s.txt")
        with open(self._imgsetpath % self.image_set) as f:
            self.ids = [x.strip("\n") for x in f.readlines()]
        self.id_to_img_map = {k: v for k, v in enumerate(self.ids)}
        cls = PascalVOCDataset.CLASSES
        self.class_to_ind = dict(zip(cls, range(len(cls))))
        self.categories = dict(zip(range(len(cls)), cls))
        self.min_size = min_size
        self.proposals = ProposalFile(proposal_file) if proposal_file is not None else None
        self.proposal_file = proposal_file
        self.top_k = 2000

    def get_origin_id(self, index):
        return self.ids[index]

    def __getitem__(self, index):
        img_id = self.ids[index]
        img = Image.open(self._imgpath % img_id).convert("RGB")
        if not os.path.exists(self._annopath % img_id):
            target = None
        else:
            target = self.get_groundtruth(index).clip_to_image(remove_empty=True)
        rois = None
        if self.proposals is not None:
            # every split keeps boxes with both sides >= 20 px (voc.py:106-109)
            rois = prepare_proposals(self.proposals.boxes(int(img_id)), img.size, min_size=20)
        if self.transforms is not None:
            img, target, rois = self.transforms(img, target, rois)
        return img, target, rois, index

    def __len__(self):
        return len(self.ids)

    def get_groundtruth(self, index):
        anno = self._preprocess_annotation(ET.parse(self._annopath % self.ids[index]).getroot())
        height, width = anno["im_info"]
        target = BoxList(anno["boxes"], (width, height), mode="xyxy")
        target.add_field("labels", anno["labels"])
        target.add_field("difficult", anno["difficult"])
        return target

    def _preprocess_annotation(self, target):
        boxes, gt_classes, difficult_boxes = [], [], []
        for obj in target.iter("object"):
            difficult = int(obj.find("difficult").text) == 1
            if not self.keep_difficult and difficult:
                continue
            name = obj.find("name").text.lower().strip()
            bb = obj.find("bndbox")
            # VOC pixel indexes are 1-based (voc.py:164-172)
            boxes.append(tuple(int(bb.find(k).text) - 1 for k in ("xmin", "ymin", "xmax", "ymax")))
            gt_classes.append(self.class_to_ind[name])
            difficult_boxes.append(difficult)
        size = target.find("size")
        return {"boxes": torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4), "labels": torch.tensor(gt_classes),
                "difficult": torch.tensor(difficult_boxes),
                "im_info": (int(size.find("height").text), int(size.find("width").text))}

    def get_img_info(self, index):
        img_id = self.ids[index]
        file_name = "JPEGImages/%s.jpg" % img_id
        if os.path.exists(self._annopath % img_id):
            size = ET.parse(self._annopath % img_id).getroot().find("size")
            return {"height": int(size.find("height").text), "width": int(size.find("width").text),
                    "file_name": file_name}
        img = Image.open(os.path.join(self.root, file_name)).convert("RGB")
        return {"height": img.size[1], "width": img.size[0], "file_name": file_name}

    def map_class_id_to_class_name(self, class_id):
        return PascalVOCDataset.CLASSES[class_id]
