from .generalized_rcnn import GeneralizedRCNN

_META = {"GeneralizedRCNN": GeneralizedRCNN}


def build_detection_model(cfg):
    """wetectron/modeling/detector/detectors.py:8."""
    return _META[cfg.MODEL.META_ARCHITECTURE](cfg)
