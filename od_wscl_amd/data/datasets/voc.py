"""PascalVOCDataset on a VOC devkit directory (JPEGImages / Annotations / ImageSets/Main) -- the dataset interface the
reference's loader, evaluator and proposal preparation expect (wetectron/data/datasets/voc.py:13-203): image-level
labels from the XML objects, proposals from a ProposalFile.  A sample is (image, target, rois, index); with transforms
the image is a `DeferredImage` (decoded pixels + pixel plan) the GPU turns into the normalised tensor at collation."""
import os
import xml.etree.ElementTree as ET

import torch
import torch.utils.data
from PIL import Image

from ...structures.bounding_box import BoxList
from .proposals import ProposalFile, prepare_proposals

VOC_CLASSES = ("__background__ ", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair",
               "cow", "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train",
               "tvmonitor")
_BOX_TAGS = ("xmin", "ymin", "xmax", "ymax")


def read_annotation(xml_path, class_index, keep_difficult):
    """One VOC XML -> (boxes (n,4) fp32 0-based xyxy, labels (n,), difficult (n,) bool, (height, width)).
    VOC pixel coordinates are 1-based (voc.py:164-172); objects flagged difficult are dropped unless asked for."""
    root = ET.parse(xml_path).getroot()
    rows = []
    for obj in root.iter("object"):
        hard = obj.find("difficult").text.strip() == "1"
        if hard and not keep_difficult:
            continue
        corner = obj.find("bndbox")
        rows.append(([int(corner.find(t).text) - 1 for t in _BOX_TAGS],
                     class_index[obj.find("name").text.lower().strip()], hard))
    dims = root.find("size")
    hw = (int(dims.find("height").text), int(dims.find("width").text))
    boxes = torch.tensor([r[0] for r in rows], dtype=torch.float32).reshape(-1, 4)
    return boxes, torch.tensor([r[1] for r in rows]), torch.tensor([r[2] for r in rows]), hw


class PascalVOCDataset(torch.utils.data.Dataset):
    CLASSES = VOC_CLASSES

    def __init__(self, data_dir, split, use_difficult=False, transforms=None, proposal_file=None, min_size=None):
        self.root, self.image_set = data_dir, split
        self.keep_difficult, self.transforms, self.min_size = use_difficult, transforms, min_size
        index_file = os.path.join(data_dir, "ImageSets", "Main", split + ".txt")
        with open(index_file) as fh:
            self.ids = [line.rstrip("\n") for line in fh]
        self.id_to_img_map = dict(enumerate(self.ids))
        self.class_to_ind = {name: k for k, name in enumerate(VOC_CLASSES)}
        self.categories = dict(enumerate(VOC_CLASSES))
        self.proposal_file = proposal_file
        self.proposals = None if proposal_file is None else ProposalFile(proposal_file)
        self.top_k = 2000

    # ---- paths
    def _xml(self, img_id):
        return os.path.join(self.root, "Annotations", img_id + ".xml")

    def _jpg(self, img_id):
        return os.path.join(self.root, "JPEGImages", img_id + ".jpg")

    # ---- the interface of the loader / evaluator
    def __len__(self):
        return len(self.ids)

    def get_origin_id(self, index):
        return self.ids[index]

    def map_class_id_to_class_name(self, class_id):
        return VOC_CLASSES[class_id]

    def get_groundtruth(self, index):
        boxes, labels, hard, (height, width) = read_annotation(self._xml(self.ids[index]), self.class_to_ind,
                                                               self.keep_difficult)
        target = BoxList(boxes, (width, height), mode="xyxy")
        target.add_field("labels", labels)
        target.add_field("difficult", hard)
        return target

    def get_img_info(self, index):
        img_id = self.ids[index]
        info = {"file_name": "JPEGImages/%s.jpg" % img_id}
        if os.path.exists(self._xml(img_id)):
            dims = ET.parse(self._xml(img_id)).getroot().find("size")
            info["height"], info["width"] = int(dims.find("height").text), int(dims.find("width").text)
        else:                                   # a test split without annotations: ask the image itself
            info["width"], info["height"] = Image.open(self._jpg(img_id)).size
        return info

    def __getitem__(self, index):
        img_id = self.ids[index]
        img = Image.open(self._jpg(img_id)).convert("RGB")
        target = None
        if os.path.exists(self._xml(img_id)):
            target = self.get_groundtruth(index).clip_to_image(remove_empty=True)
        rois = None
        if self.proposals is not None:          # every split keeps boxes with both sides >= 20 px (voc.py:106-109)
            rois = prepare_proposals(self.proposals.boxes(int(img_id)), img.size, min_size=20)
        if self.transforms is not None:
            img, target, rois = self.transforms(img, target, rois)
        return img, target, rois, index
