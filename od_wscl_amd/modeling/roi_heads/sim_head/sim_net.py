"""Sim_Net (wetectron/modeling/roi_heads/sim_head/sim_net.py:7-26): 4096 -> 4096 -> 128,
L2-normalised rows."""
import torch.nn as nn
import torch.nn.functional as F


class Sim_Net(nn.Module):
    def __init__(self, config, in_dim):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(in_dim, in_dim), nn.ReLU(inplace=True), nn.Linear(in_dim, 128))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)

    def forward(self, roi_feat):
        return F.normalize(self.mlp(roi_feat), dim=1)
