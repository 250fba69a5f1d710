"""VGG16-OICR backbone and the fc6/fc7 ROI feature extractor
(wetectron/modeling/backbone/vgg16.py:26-193).

Parameter names match the reference's state-dict (`body.features.N`, `classifier.{1,4}`)
so reference checkpoints load unchanged."""
from collections import OrderedDict

import torch.nn as nn

from .. import registry
from ...layers import Conv2d
from ...layers.linear import Linear
from .fc_extractor import TwoFCROIFeatureExtractor

# conv widths, 'M' = 2x2 max-pool, 'I' = identity (pool4 removed), '-D' = dilation 2 (vgg16.py:86-93)
VGG_CFG = {
    "VGG16": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512],
    "VGG16-OICR": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "I", "512-D", "512-D", "512-D"],
}


def make_layers(spec):
    layers, cin = [], 3
    for v in spec:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        elif v == "I":
            layers.append(nn.Identity())
        else:
            dil = 2 if isinstance(v, str) else 1
            cout = int(str(v).split("-")[0])
            layers += [Conv2d(cin, cout, kernel_size=3, padding=dil, dilation=dil), nn.ReLU(inplace=True)]
            cin = cout
    return nn.Sequential(*layers[:-1])      # the last ReLU is dropped (vgg16.py:82-83, Q7)


class VGG_Base(nn.Module):
    def __init__(self, features, cfg, init_weights=True):
        super().__init__()
        self.features = features
        if init_weights:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                    nn.init.constant_(m.bias, 0)
        self._freeze_backbone(cfg.MODEL.BACKBONE.FREEZE_CONV_BODY_AT)

    def forward(self, x):
        return [self.features(x)]

    def _freeze_backbone(self, freeze_at):
        if freeze_at < 0:
            return
        upto = [5, 10, 17, 23, 29][freeze_at - 1]      # vgg16.py:48-55
        for layer in list(self.features)[:upto]:
            for p in layer.parameters():
                p.requires_grad = False


@registry.BACKBONES.register("VGG16")
@registry.BACKBONES.register("VGG16-OICR")
def add_conv_body(cfg, dim_in=3):
    body = VGG_Base(make_layers(VGG_CFG[cfg.MODEL.BACKBONE.CONV_BODY]), cfg)
    model = nn.Sequential(OrderedDict([("body", body)]))
    model.out_channels = 512
    return model


def build_backbone(cfg):
    name = cfg.MODEL.BACKBONE.CONV_BODY
    if name not in registry.BACKBONES:
        raise KeyError("cfg.MODEL.BACKBONE.CONV_BODY: %s is not registered" % name)
    return registry.BACKBONES[name](cfg)


@registry.ROI_BOX_FEATURE_EXTRACTORS.register("VGG16.roi_head")
class VGG16FC67ROIFeatureExtractor(TwoFCROIFeatureExtractor):
    """Pooler -> flatten -> fc6, ReLU, Dropout, fc7, ReLU, Dropout (vgg16.py:107-180); the shared
    machinery (stacked clean+DropBlock pass, noise / drop views) lives in fc_extractor.py."""

    def __init__(self, config, in_channels, init_weights=True):
        super().__init__(config)
        assert in_channels == 512
        res = config.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        self.classifier = nn.Sequential(nn.Identity(), Linear(512 * res * res, 4096), nn.ReLU(inplace=True),
                                        nn.Dropout(), Linear(4096, 4096), nn.ReLU(inplace=True), nn.Dropout())
        self.fc_index = (1, 4)
        self.fc6.cm_layout = (512, res * res)      # fc6 reduces over (channel, cell): what the shared clean + DropBlock forward walks
        self.out_channels = 4096
        if init_weights:
            self.init_fc()
