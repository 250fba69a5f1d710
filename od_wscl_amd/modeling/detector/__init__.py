from .detectors import build_detection_model
from .generalized_rcnn import GeneralizedRCNN

__all__ = ["build_detection_model", "GeneralizedRCNN"]
