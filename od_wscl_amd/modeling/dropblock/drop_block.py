"""DropBlock2D (wetectron/modeling/dropblock/drop_block.py:7-71).

One Bernoulli(drop_prob / block_size^2) mask per ROI over the 7x7 grid, shared by all
channels, dilated by a block_size max-pool, applied and rescaled by numel/sum over the
WHOLE mask tensor.  The reference draws the mask with the CPU generator and copies it to
the device (:42-44); here the draw is the counter-based device stream, the (N,7,7) mask
algebra stays in tiny tensors and the big (N,C,7,7) tensor is touched once."""
import torch
import torch.nn.functional as F
from torch import nn


class DropBlock2D(nn.Module):
    def __init__(self, drop_prob, block_size):
        super().__init__()
        self.drop_prob = drop_prob
        self.block_size = block_size

    def keep_mask(self, n, h, w, device, rand=None):
        """The (n, h, w) keep mask after dilation (drop_block.py:38-47, :55-71) -- one draw of n*h*w uniforms."""
        gamma = self.drop_prob / (self.block_size ** 2)
        shape = (n, h, w)
        if (rand is not None and hasattr(rand, "key") and torch.device(device).type == "cuda" and n * h * w < (1 << 24)
                and self.block_size <= 15):         # (the kernel re-derives a cell's block_size^2 draws: <= 15 x 15; larger blocks
                                                    # take the torch expression below, any size)
            # the counter-based device stream: draw, threshold, dilation, inversion and the sum in ONE kernel
            # (csrc/rng.hip dropblock_keep_kernel) -- the same draw, the same stream id as rand.uniform(shape)
            from ... import _lib as L
            k0, k1 = rand.key()
            keep = torch.empty(shape, dtype=torch.float32, device=device)
            total = torch.empty((), dtype=torch.float32, device=device)
            L.check(L.lib().odw_dropblock_keep_mask(n, h, w, int(self.block_size), float(gamma), k0, k1, L.ptr(keep), L.ptr(total),
                                                    L.stream()), "dropblock_keep_mask")
            keep._odw_sum = total           # callers that need block.sum() take this instead of a reduction launch
            return keep
        u = rand.uniform(shape) if rand is not None else torch.rand(shape, device=device)
        centres = (u < gamma).float()
        block = F.max_pool2d(centres[:, None], kernel_size=self.block_size, stride=1, padding=self.block_size // 2)
        if self.block_size % 2 == 0:
            block = block[:, :, :-1, :-1]
        return 1 - block.squeeze(1)

    def forward(self, x, rand=None):
        assert x.dim() == 4, "Expected input with 4 dimensions (bsize, channels, height, width)"
        if not self.training or self.drop_prob == 0.0:
            return x
        block = self.keep_mask(x.shape[0], x.shape[2], x.shape[3], x.device, rand)
        # (x * block) * numel / sum, evaluated in the reference's order (:49-50)
        total = getattr(block, "_odw_sum", None)
        return x * block[:, None, :, :] * block.numel() / (total if total is not None else block.sum())
