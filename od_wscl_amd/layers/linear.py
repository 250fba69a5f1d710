"""Linear layer of the ROI head (fc6 / fc7 / Sim_Net / predictor; the 1x1 convolutions of the ResNet bodies).

Keeps nn.Linear's parameters (state-dict names unchanged).  Every product runs on the hand-written gfx950 MFMA GEMM
with its fused bias / ReLU / dropout epilogue (csrc/gemm_bf16.hip); there is no library or CPU path.  The arithmetic
precision is the process-wide setting of od_wscl_amd.precision: "bf16" (throughput) or "bf16x3" (fp32-grade: the
reference's DTYPE float32, config/defaults.py:559, on the bf16 matrix cores by operand splitting).
`tag` names the layer for the bench's per-kernel timing."""
import os

from torch import nn

from .. import gemm


class Linear(nn.Linear):
    tag = None
    _shadow = None

    def fused(self, x, relu=False, drop_p=0.0, key=None, segs=None, out_f32=False, grad_rows=None, row_ids=None):
        """dropout(relu(x W^T + b)); `key` = (k0,k1) of one counter-based draw or `segs` =
        [(first_row, k0, k1), ...] when several logical passes are stacked along M."""
        if not x.is_cuda:
            raise RuntimeError("od_wscl_amd.layers.Linear: tensor is not on the GPU -- the hot path has no CPU "
                               "implementation (the MFMA GEMM of libodwscl.so is the only one)")
        self._get_shadow()
        if drop_p > 0 and segs is None:
            segs = [(0, key[0], key[1])]
        return gemm.fused_linear(x, self.weight, self.bias, self._shadow, relu=relu, drop_p=drop_p,
                                 segs=segs if drop_p > 0 else None, out_f32=out_f32, tag=self.tag,
                                 grad_rows=grad_rows, row_ids=row_ids)

    cm_layout = None       # (C, S): this layer reduces over a (C, S-cell) map flattened channel-major (the first head Linear)

    def _get_shadow(self):
        """The bf16 copies of the weight (gemm.Shadow), created on first use; a layer with a cm_layout keeps its forward
        operand as cell-major planes in the precision mode "bf16x2f" (ODW_NO_PAIR=1: the channel-major planes of the
        other layers)."""
        if self._shadow is None or self._shadow.weight is not self.weight:
            self._shadow = gemm.Shadow(self.weight)
            if (self.cm_layout is not None and os.environ.get("ODW_NO_PAIR") != "1"
                    and self.cm_layout[0] % 64 == 0 and 1 <= self.cm_layout[1] <= 64):      # (what gemm_nt_cm_kernel walks)
                self._shadow.cm = tuple(self.cm_layout)
        return self._shadow

    def can_pair(self, C, S):
        """The shared clean + DropBlock forward applies: the layer reduces over (C, S) and -- when an optimiser keeps its
        bf16 copies (engine.FlatSGD) -- that optimiser also keeps the cell-major planes."""
        if self.cm_layout is None or tuple(self.cm_layout) != (C, S) or C % 64 != 0 or S > 64:
            return False
        sh = self._get_shadow()
        return sh.cm is not None and (not sh.managed or sh.w_cm is not None)

    def pair(self, x, planes_cm, planes_bwd, keep, keep_sum, relu=False, drop_p=0.0, segs=None, grad_rows=None):
        """dropout(relu(.)) of the clean rows AND of their DropBlock view from one sweep over the clean operand
        (gemm.pair_linear); x = the autograd handle of the stacked (2P x K) operand."""
        self._get_shadow()
        return gemm.pair_linear(x, self.weight, self.bias, self._shadow, planes_cm, planes_bwd, keep, keep_sum, relu=relu,
                                drop_p=drop_p, segs=segs, tag=self.tag, grad_rows=grad_rows)

    def reuse(self, x, y_full, rows, relu=False, drop_p=0.0):
        """Rows `rows` of an earlier no-autograd evaluation `y_full` of this layer, re-attached to the graph with `x` as
        their input (gemm.reuse_linear): backward as usual, no forward GEMM."""
        self._get_shadow()
        return gemm.reuse_linear(x, self.weight, self.bias, self._shadow, y_full, rows, relu=relu, drop_p=drop_p, tag=self.tag)

    def forward(self, x):
        return self.fused(x)
