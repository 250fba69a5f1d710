"""Stand-in for pycocotools.coco (absent here): the index part of its published COCO class -- imgs / anns / cats
dictionaries from the annotation json, getAnnIds(imgIds, iscrowd), loadAnns, getCatIds (sorted), loadImgs -- enough
for the reference's COCODataset.  Test infrastructure only."""
import json
from collections import defaultdict


class COCO(object):
    def __init__(self, annotation_file=None):
        self.dataset, self.anns, self.cats, self.imgs = {}, {}, {}, {}
        self.imgToAnns = defaultdict(list)
        if annotation_file is not None:
            with open(annotation_file) as f:
                self.dataset = json.load(f)
            for ann in self.dataset.get("annotations", []):
                self.imgToAnns[ann["image_id"]].append(ann)
                self.anns[ann["id"]] = ann
            for img in self.dataset.get("images", []):
                self.imgs[img["id"]] = img
            for cat in self.dataset.get("categories", []):
                self.cats[cat["id"]] = cat

    def getAnnIds(self, imgIds=(), iscrowd=None):
        imgIds = imgIds if isinstance(imgIds, (list, tuple)) else [imgIds]
        anns = [a for i in imgIds for a in self.imgToAnns.get(i, [])] if len(imgIds) else list(self.dataset.get("annotations", []))
        if iscrowd is not None:
            anns = [a for a in anns if a["iscrowd"] == iscrowd]
        return [a["id"] for a in anns]

    def loadAnns(self, ids=()):
        ids = ids if isinstance(ids, (list, tuple)) else [ids]
        return [self.anns[i] for i in ids]

    def getCatIds(self):
        return sorted(self.cats.keys())

    def loadImgs(self, ids=()):
        ids = ids if isinstance(ids, (list, tuple)) else [ids]
        return [self.imgs[i] for i in ids]
