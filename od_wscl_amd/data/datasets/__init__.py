from .voc import PascalVOCDataset  # noqa: F401
from .coco import COCODataset  # noqa: F401
from .concat_dataset import ConcatDataset  # noqa: F401
from .proposals import ProposalFile, prepare_proposals, unique_boxes  # noqa: F401

__all__ = ["COCODataset", "ConcatDataset", "PascalVOCDataset", "ProposalFile"]
