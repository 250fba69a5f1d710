"""ResNet-C5 backbones of the hot path (R-50-C5 / R-101-C5; wetectron/modeling/backbone/resnet.py:84-148,
:228-252, :258-375, :381-399 and backbone.py:14-23) and the two-Linear ROI feature extractor that goes with them
(roi_heads/box_head/roi_box_feature_extractors.py:13-122).

Only what the OD-WSCL configs instantiate is built: bottleneck blocks with frozen batch-norm
(TRANS_FUNC BottleneckWithFixedBatchNorm, STEM_FUNC StemWithFixedBatchNorm), no groups-norm, no deformable
convolutions, no FPN.  Parameter and buffer names (`body.stem.conv1`, `body.layerN.M.{conv,bn}K`,
`body.layerN.0.downsample.{0,1}`, `classifier.{0,3}`) match the reference's state-dict so MSRA R-50 checkpoints
and reference checkpoints load unchanged."""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import registry
from ...layers import Conv2d, FrozenBatchNorm2d
from .fc_extractor import TwoFCROIFeatureExtractor

# blocks per stage (layer1..layer4); every *-C5 body returns layer4 only (resnet.py:35-65)
STAGE_BLOCKS = {"R-50-C5": (3, 4, 6, 3), "R-101-C5": (3, 4, 23, 3)}


def _conv(cin, cout, k, stride=1, padding=0, dilation=1, groups=1):
    c = Conv2d(cin, cout, kernel_size=k, stride=stride, padding=padding, dilation=dilation, groups=groups, bias=False)
    nn.init.kaiming_uniform_(c.weight, a=1)                  # resnet.py:289,333,342,396
    return c


class Stem(nn.Module):
    """7x7/2 conv, frozen BN, ReLU, 3x3/2 max-pool (resnet.py:381-406)."""

    def __init__(self, cfg):
        super().__init__()
        c = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
        self.conv1 = _conv(3, c, 7, stride=2, padding=3)
        self.bn1 = FrozenBatchNorm2d(c)

    def forward(self, x):
        x = F.relu_(self.bn1(self.conv1(x)))
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 -> 1x1 with frozen BN after each and a projected shortcut when the width changes
    (resnet.py:258-375).  With STRIDE_IN_1X1 (MSRA weights) the stride sits on the first 1x1."""

    def __init__(self, cin, cmid, cout, groups, stride_in_1x1, stride, dilation=1):
        super().__init__()
        self.downsample = None
        if cin != cout:
            self.downsample = nn.Sequential(_conv(cin, cout, 1, stride=stride if dilation == 1 else 1),
                                            FrozenBatchNorm2d(cout))
        if dilation > 1:
            stride = 1
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = _conv(cin, cmid, 1, stride=s1)
        self.bn1 = FrozenBatchNorm2d(cmid)
        self.conv2 = _conv(cmid, cmid, 3, stride=s3, padding=dilation, dilation=dilation, groups=groups)
        self.bn2 = FrozenBatchNorm2d(cmid)
        self.conv3 = _conv(cmid, cout, 1)
        self.bn3 = FrozenBatchNorm2d(cout)

    def forward(self, x):
        y = F.relu_(self.bn1(self.conv1(x)))
        y = F.relu_(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y = y + (x if self.downsample is None else self.downsample(x))
        return F.relu_(y)


class ResNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        r = cfg.MODEL.RESNETS
        if r.TRANS_FUNC != "BottleneckWithFixedBatchNorm" or r.STEM_FUNC != "StemWithFixedBatchNorm":
            raise NotImplementedError("only the frozen-batch-norm ResNets are on the OD-WSCL hot path")
        if any(r.STAGE_WITH_DCN):
            raise NotImplementedError("deformable convolutions are outside the OD-WSCL hot path")
        self.stem = Stem(cfg)
        cin, mid0, out0 = r.STEM_OUT_CHANNELS, r.NUM_GROUPS * r.WIDTH_PER_GROUP, r.RES2_OUT_CHANNELS
        self.stages = []
        for i, count in enumerate(STAGE_BLOCKS[cfg.MODEL.BACKBONE.CONV_BODY]):
            mid, out = mid0 << i, out0 << i
            blocks = [Bottleneck(cin if b == 0 else out, mid, out, r.NUM_GROUPS, r.STRIDE_IN_1X1,
                                 (1 + int(i > 0)) if b == 0 else 1) for b in range(count)]
            self.add_module("layer%d" % (i + 1), nn.Sequential(*blocks))
            self.stages.append("layer%d" % (i + 1))
            cin = out
        self.out_channels = cin
        self._freeze_backbone(cfg.MODEL.BACKBONE.FREEZE_CONV_BODY_AT)

    def _freeze_backbone(self, freeze_at):
        """stage 0 = stem, stage k = layerk (resnet.py:128-137)."""
        for k in range(max(freeze_at, 0)):
            m = self.stem if k == 0 else getattr(self, "layer%d" % k)
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, x):
        x = self.stem(x)
        for name in self.stages:
            x = getattr(self, name)(x)
        return [x]


@registry.BACKBONES.register("R-50-C5")
@registry.BACKBONES.register("R-101-C5")
def build_resnet_backbone(cfg):
    """backbone.py:14-23 plus the stride patch GeneralizedRCNN applies to every *-C5 body
    (detector/generalized_rcnn.py:37-45): layer4 runs at stride 16, not 32, so POOLER_SCALES is 1/16.

    `out_channels` is cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS as in the reference (1024 by default although
    layer4 emits 2048 channels -- the extractor below ignores it and hard-codes 7*7*2048, Q13)."""
    body = ResNet(cfg)
    first = body.layer4[0]
    first.downsample[0].stride = (1, 1)
    first.conv1.stride = (1, 1)
    model = nn.Sequential(OrderedDict([("body", body)]))
    model.out_channels = cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS
    return model


@registry.ROI_BOX_FEATURE_EXTRACTORS.register("ResNet50Conv5ROIFeatureExtractor")
@registry.ROI_BOX_FEATURE_EXTRACTORS.register("ResNet101Conv5ROIFeatureExtractor")
class ResNet50Conv5ROIFeatureExtractor(TwoFCROIFeatureExtractor):
    """Pooler -> flatten -> Linear(7*7*2048, 2048), ReLU, Dropout, Linear(2048, 4096), ReLU, Dropout
    (roi_box_feature_extractors.py:13-122; the res5 head of the upstream maskrcnn-benchmark class is commented
    out there).  The input width is the literal 7*7*2048 whatever POOLER_RESOLUTION / in_channels say."""

    def __init__(self, config, in_channels, init_weights=True):
        super().__init__(config)
        self.classifier = nn.Sequential(self.Linear(7 * 7 * 2048, 2048), nn.ReLU(inplace=True), nn.Dropout(),
                                        self.Linear(2048, 4096), nn.ReLU(inplace=True), nn.Dropout())
        self.fc_index = (0, 3)
        # fc6 reduces over (channel, cell) of a 2048 x 7 x 7 map: the shared clean + DropBlock forward of "bf16x2f" walks it cell
        # by cell (gemm.pair_linear: half the matrix-core work of the stacked pass), as for VGG16's fc6 (vgg16.py)
        self.fc6.cm_layout = (2048, 7 * 7)
        self.out_channels = 4096
        if init_weights:
            self.init_fc()
