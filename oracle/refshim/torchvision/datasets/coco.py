"""Stand-in for torchvision.datasets.coco.CocoDetection (torchvision 0.8.2): ids = sorted image ids of the annotation
file, __getitem__ -> (PIL RGB image, list of annotation dicts of that image).  Test infrastructure only."""
import os

from PIL import Image


class CocoDetection(object):
    def __init__(self, root, annFile, transform=None, target_transform=None, transforms=None):
        from pycocotools.coco import COCO
        self.root = root
        self.coco = COCO(annFile)
        self.ids = list(sorted(self.coco.imgs.keys()))

    def __getitem__(self, index):
        img_id = self.ids[index]
        target = self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id))
        path = self.coco.loadImgs(img_id)[0]["file_name"]
        img = Image.open(os.path.join(self.root, path)).convert("RGB")
        return img, target

    def __len__(self):
        return len(self.ids)
