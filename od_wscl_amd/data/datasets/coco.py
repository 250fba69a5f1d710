"""COCODataset for the detection configs (wetectron/data/datasets/coco.py:21-45,60-121,186-196) without pycocotools:
the instances json is indexed directly.  Sorted image ids, optional removal of images without a usable annotation
(all boxes with a side <= 1 px count as none), crowd objects dropped, json category ids mapped to contiguous 1..80,
xywh boxes -> xyxy (+1 convention) clipped to the image, proposals from the reference's pickle with duplicates /
empty / tiny (side < 2 px) boxes removed.  Masks, keypoints, clicks and scribbles are outside the OD-WSCL path."""
import json
import os
from collections import defaultdict

import torch
import torch.utils.data
from PIL import Image

from ...structures.bounding_box import BoxList
from .proposals import ProposalFile, prepare_proposals


def has_valid_annotation(anno):
    if len(anno) == 0:
        return False
    if all(any(o <= 1 for o in obj["bbox"][2:]) for obj in anno):
        return False
    return True


class COCODataset(torch.utils.data.Dataset):
    def __init__(self, ann_file, root, remove_images_without_annotations, transforms=None, proposal_file=None,
                 min_size=None):
        with open(ann_file) as f:
            data = json.load(f)
        self.root, self.ann_file, self.min_size = root, ann_file, min_size
        self.imgs = {im["id"]: im for im in data.get("images", [])}
        self.img_to_anns = defaultdict(list)
        for ann in data.get("annotations", []):
            self.img_to_anns[ann["image_id"]].append(ann)
        cats = {c["id"]: c for c in data.get("categories", [])}
        self.ids = sorted(self.imgs.keys())
        if remove_images_without_annotations:
            self.ids = [i for i in self.ids if has_valid_annotation(self.img_to_anns.get(i, []))]
        self.categories = {c["id"]: c["name"] for c in cats.values()}
        self.json_category_id_to_contiguous_id = {v: i + 1 for i, v in enumerate(sorted(cats.keys()))}
        self.contiguous_category_id_to_json_id = {v: k for k, v in self.json_category_id_to_contiguous_id.items()}
        self.id_to_img_map = {k: v for k, v in enumerate(self.ids)}
        self._transforms = transforms
        self.proposals = ProposalFile(proposal_file) if proposal_file is not None else None

    def __len__(self):
        return len(self.ids)

    def _annotations(self, idx):
        return self.img_to_anns.get(self.ids[idx], [])

    def __getitem__(self, idx):
        img_id = self.ids[idx]
        img = Image.open(os.path.join(self.root, self.imgs[img_id]["file_name"])).convert("RGB")
        anno = self._annotations(idx)
        if "lvis_v0.5" not in self.ann_file:
            anno = [obj for obj in anno if obj["iscrowd"] == 0]
        rois = None
        if self.proposals is not None:
            rois = prepare_proposals(self.proposals.boxes(img_id), img.size, min_size=2)     # coco.py:117
        if anno == [] and "unlabeled" in self.ann_file:
            target = BoxList(torch.zeros((1, 4)), img.size, mode="xyxy")
            target.add_field("labels", torch.tensor([0]))
            if self._transforms is not None:
                img, target, rois = self._transforms(img, target, rois)
            target.bbox.fill_(0)
            return img, target, rois, idx
        boxes = torch.as_tensor([obj["bbox"] for obj in anno], dtype=torch.float32).reshape(-1, 4)
        target = BoxList(boxes, img.size, mode="xywh").convert("xyxy")
        target.add_field("labels", torch.tensor([self.json_category_id_to_contiguous_id[obj["category_id"]] for obj in anno]))
        target = target.clip_to_image(remove_empty=True)
        if self._transforms is not None:
            img, target, rois = self._transforms(img, target, rois)
        return img, target, rois, idx

    def get_img_info(self, index):
        return self.imgs[self.id_to_img_map[index]]

    def get_groundtruth(self, index):
        """Contiguous class ids of the image's annotations, crowd included (coco.py:191-196) -- a plain tensor."""
        return torch.tensor([self.json_category_id_to_contiguous_id[obj["category_id"]] for obj in self._annotations(index)])
