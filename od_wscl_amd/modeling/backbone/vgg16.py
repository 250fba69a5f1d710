"""VGG16-OICR backbone and the fc6/fc7 ROI feature extractor
(wetectron/modeling/backbone/vgg16.py:26-193).

Parameter names match the reference's state-dict (`body.features.N`, `classifier.{1,4}`)
so reference checkpoints load unchanged."""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import registry
from ..dropblock import DropBlock2D
from ..poolers import Pooler
from ...layers import Conv2d
from ...layers.linear import Linear

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
class VGG16FC67ROIFeatureExtractor(nn.Module):
    """Pooler -> flatten -> fc6, ReLU, Dropout, fc7, ReLU, Dropout, plus the DropBlock / noise
    views the contrastive loss asks for (vgg16.py:107-180).  `rand` (a DeviceRand) carries the
    counter-based streams; every method draws in the reference's order."""

    def __init__(self, config, in_channels, init_weights=True):
        super().__init__()
        assert in_channels == 512
        res = config.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        self.pooler = Pooler(output_size=(res, res), scales=config.MODEL.ROI_BOX_HEAD.POOLER_SCALES,
                             sampling_ratio=config.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO,
                             method=config.MODEL.ROI_BOX_HEAD.POOLER_METHOD)
        self.classifier = nn.Sequential(nn.Identity(), Linear(512 * res * res, 4096), nn.ReLU(inplace=True),
                                        nn.Dropout(), Linear(4096, 4096), nn.ReLU(inplace=True), nn.Dropout())
        self.out_channels = 4096
        if config.DB.METHOD == "dropblock":
            self.dropblock = DropBlock2D(block_size=3, drop_prob=0.3)
        self.sim_drop = DropBlock2D(block_size=1, drop_prob=0.3)
        self.rand = None
        if init_weights:
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.normal_(m.weight, 0, 0.01)
                    nn.init.constant_(m.bias, 0)

    def _fc(self, x, segs6=None, segs7=None):
        """fc6, ReLU, Dropout, fc7, ReLU, Dropout (vgg16.py:121-127).  With a counter-based `rand`
        the two dropouts are fused into the GEMM epilogues; `segs*` carry per-pass keys when
        several passes are stacked along the row dimension."""
        fc6, fc7 = self.classifier[1], self.classifier[4]
        if not self.training:
            return torch.relu(fc7(torch.relu(fc6(x))))
        if self.rand is None:
            x = F.dropout(torch.relu(fc6(x)), 0.5, True)
            return F.dropout(torch.relu(fc7(x)), 0.5, True)
        if segs6 is None:
            k6, k7 = self.rand.key(), self.rand.key()
            segs6, segs7 = [(0, k6[0], k6[1])], [(0, k7[0], k7[1])]
        x = fc6.fused(x, relu=True, drop_p=0.5, segs=segs6)
        return fc7.fused(x, relu=True, drop_p=0.5, segs=segs7)

    def forward_clean_and_aug(self, pooled):
        """The clean pass and the DropBlock pass of ROIWeakRegHead.forward (weak_head.py:107-112) as
        ONE stacked fc6/fc7 evaluation (M = 2P): random draws are numbered in the reference's
        order (clean fc6, clean fc7, DropBlock centres, aug fc6, aug fc7)."""
        P = pooled.shape[0]
        k1, k2 = self.rand.key(), self.rand.key()
        aug = self.forward_dropblock(pooled) if hasattr(self, "dropblock") else pooled
        k4, k5 = self.rand.key(), self.rand.key()
        x = torch.cat([pooled.reshape(P, -1), aug.reshape(P, -1)], dim=0)
        h = self._fc(x, segs6=[(0,) + k1, (P,) + k4], segs7=[(0,) + k2, (P,) + k5])
        return h[:P], h[P:]

    def forward(self, x, proposals):
        pooled = self.pooler(x, proposals)
        return self._fc(pooled.reshape(pooled.shape[0], -1)), pooled

    def forward_pooler(self, x, proposals):
        return self.pooler(x, proposals)

    def forward_neck(self, x):
        return self._fc(x.reshape(x.shape[0], -1))

    def forward_dropblock(self, pooled_feats, proposals=None):
        return self.dropblock(pooled_feats, self.rand)

    def drop_pool(self, pooled_feats):
        return self.sim_drop(pooled_feats, self.rand)

    def noise_pool(self, pooled_feats):
        if self.rand is not None:
            return self.rand.noise_mul(pooled_feats)
        noise = torch.randn_like(pooled_feats)
        return noise * pooled_feats + pooled_feats
