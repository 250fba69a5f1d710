"""BatchCollator / BBoxAugCollator (wetectron/data/collate_batch.py:5-39).  The image part of a collated batch is a
`PendingImageList` (decoded pixels + plans); it becomes the padded fp32 batch on the GPU in `.to(device)`."""
from ..structures.image_list import to_image_list


class BatchCollator(object):
    def __init__(self, size_divisible=0):
        self.size_divisible = size_divisible

    def __call__(self, batch):
        transposed_batch = list(zip(*batch))
        images = to_image_list(list(transposed_batch[0]), self.size_divisible)
        targets = transposed_batch[1]
        if len(transposed_batch) == 3:
            return images, targets, transposed_batch[2]
        if len(transposed_batch) == 4:
            return images, targets, transposed_batch[2], transposed_batch[3]
        raise ValueError("wrong item")


class BBoxAugCollator(object):
    """Test-time augmentation: the raw images travel as they are, `im_detect_bbox_aug` runs the transforms."""

    def __call__(self, batch):
        return list(zip(*batch))
