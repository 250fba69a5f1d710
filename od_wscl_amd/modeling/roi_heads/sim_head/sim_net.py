"""Sim_Net (wetectron/modeling/roi_heads/sim_head/sim_net.py:7-26): 4096 -> 4096 -> 128,
L2-normalised rows.  On the gfx950 back end both Linears run on the MFMA GEMM (ReLU fused into
the first epilogue, fp32 output from the second so the normalisation sees full precision)."""
import torch.nn as nn
import torch.nn.functional as F

from ....layers.linear import Linear, get_backend


class Sim_Net(nn.Module):
    def __init__(self, config, in_dim):
        super().__init__()
        self.mlp = nn.Sequential(Linear(in_dim, in_dim), nn.ReLU(inplace=True), Linear(in_dim, 128))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)

    def forward(self, roi_feat):
        if get_backend() == "hip_bf16":
            h = self.mlp[0].fused(roi_feat, relu=True)
            return F.normalize(self.mlp[2].fused(h, out_f32=True), dim=1)
        return F.normalize(self.mlp(roi_feat), dim=1)
