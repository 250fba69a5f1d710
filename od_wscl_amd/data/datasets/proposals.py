"""Proposal files of the reference: a pickle of `{boxes: [int16 (n,4) xyxy per image], scores: [fp32 (n)],
indexes (or ids): [image id]}` written by wetectron/utils/proposal_convert.py:45-51,96-109 (MCG / selective-search
boxes, 0-based, x1 y1 x2 y2), and what the datasets do to one image's boxes before the transforms see them
(data/datasets/voc.py:94-111, coco.py:52-57,104-121): drop duplicates, clip to the image, drop empty boxes, drop boxes
with a side under 20 px.  Host-side numpy; ~2-4 k boxes per image."""
import pickle

import numpy as np
import torch

from ...structures.bounding_box import BoxList


def unique_boxes(boxes, scale=1.0):
    """Indices of the first occurrence of every distinct box, ascending (coco.py:52-57)."""
    v = np.array([1, 1e3, 1e6, 1e9])
    hashes = np.round(boxes * scale).dot(v)
    _, index = np.unique(hashes, return_index=True)
    return np.sort(index)


def prepare_proposals(raw_boxes, image_size, min_size=20):
    """int16 (n,4) boxes of one image -> BoxList of the proposals the model sees, image_size = (w, h)."""
    raw_boxes = np.asarray(raw_boxes)
    b = raw_boxes[unique_boxes(raw_boxes)].astype(np.float64).astype(np.float32)
    w, h = image_size
    np.clip(b[:, 0::2], 0, w - 1, out=b[:, 0::2])
    np.clip(b[:, 1::2], 0, h - 1, out=b[:, 1::2])
    b = b[(b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])]
    keep = (b[:, 2] - b[:, 0] + 1 >= min_size) & (b[:, 3] - b[:, 1] + 1 >= min_size)
    return BoxList(torch.from_numpy(np.ascontiguousarray(b[keep])), image_size, mode="xyxy")


class ProposalFile(object):
    """The loaded pickle + an id -> position map (the reference runs `list.index` per sample, voc.py:101)."""

    def __init__(self, path):
        with open(path, "rb") as f:
            self.data = pickle.load(f, encoding="latin1")
        self.id_field = "indexes" if "indexes" in self.data else "ids"
        self.position = {}
        for k, v in enumerate(self.data[self.id_field]):
            self.position.setdefault(v, k)          # first match, like list.index

    def boxes(self, image_id):
        if image_id not in self.position:
            raise ValueError("%r is not in list" % (image_id,))
        return self.data["boxes"][self.position[image_id]]

    @staticmethod
    def write(path, boxes, scores, indexes):
        """The layout proposal_convert.py dumps (:50-51,:108-109)."""
        with open(path, "wb") as f:
            pickle.dump(dict(boxes=[np.asarray(b).astype(np.int16) for b in boxes],
                             scores=[np.squeeze(np.asarray(s).astype(np.float32)) for s in scores],
                             indexes=list(indexes)), f, pickle.HIGHEST_PROTOCOL)
