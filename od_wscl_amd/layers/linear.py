"""Linear layer of the ROI head (fc6 / fc7 / Sim_Net / predictor GEMMs).

Keeps nn.Linear's parameters (state-dict names unchanged).  Two back ends:
  "hip_bf16" : the hand-written gfx950 MFMA GEMM with fused bias/ReLU/dropout epilogue
               (csrc/gemm_bf16.hip) -- the production path
  "torch"    : F.linear in the tensor's dtype (fp32 = the parity mode that reproduces the
               reference's numerics to 1e-6; comparison baseline for the bench)
`tag` names the layer for the bench's per-kernel timing."""
import torch
import torch.nn.functional as F
from torch import nn

from ..utils.kernel_timer import kernel_timer

_BACKEND = "torch"


def set_backend(name):
    global _BACKEND
    if name not in ("torch", "hip_bf16"):
        raise ValueError(name)
    _BACKEND = name


def get_backend():
    return _BACKEND


class Linear(nn.Linear):
    tag = None
    _shadow = None

    def fused(self, x, relu=False, drop_p=0.0, key=None, segs=None, out_f32=False, grad_rows=None, row_ids=None):
        """dropout(relu(x W^T + b)); `key` = (k0,k1) of one counter-based draw or `segs` =
        [(first_row, k0, k1), ...] when several logical passes are stacked along M."""
        if _BACKEND == "hip_bf16":
            from .. import gemm
            if self._shadow is None or self._shadow.weight is not self.weight:
                self._shadow = gemm.Shadow(self.weight)
            if drop_p > 0 and segs is None:
                segs = [(0, key[0], key[1])]
            return gemm.fused_linear(x, self.weight, self.bias, self._shadow, relu=relu, drop_p=drop_p,
                                     segs=segs if drop_p > 0 else None, out_f32=out_f32, tag=self.tag,
                                     grad_rows=grad_rows, row_ids=row_ids)
        y = self.forward(x)
        if relu:
            y = torch.relu(y)
        if drop_p > 0:
            from ..utils.device_rand import dropout_with_segments
            y = dropout_with_segments(y, drop_p, segs if segs is not None else [(0, key[0], key[1])])
        return y

    def forward(self, x):
        if self.tag is None or not kernel_timer.enabled:
            return F.linear(x, self.weight, self.bias)
        flops = 2.0 * x.shape[0] * self.in_features * self.out_features
        with kernel_timer.region("layer/" + self.tag + "_fwd", flops=flops):
            return F.linear(x, self.weight, self.bias)
