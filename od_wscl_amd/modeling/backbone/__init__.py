from .vgg16 import VGG_Base, VGG16FC67ROIFeatureExtractor, add_conv_body, build_backbone
from .resnet import ResNet, ResNet50Conv5ROIFeatureExtractor, build_resnet_backbone

__all__ = ["VGG_Base", "VGG16FC67ROIFeatureExtractor", "add_conv_body", "build_backbone",
           "ResNet", "ResNet50Conv5ROIFeatureExtractor", "build_resnet_backbone"]
