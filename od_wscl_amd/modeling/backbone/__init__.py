from .vgg16 import VGG_Base, VGG16FC67ROIFeatureExtractor, add_conv_body, build_backbone

__all__ = ["VGG_Base", "VGG16FC67ROIFeatureExtractor", "add_conv_body", "build_backbone"]
