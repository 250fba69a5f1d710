"""smooth_l1_loss with the extra `beta`, `size_average`, `reduction` arguments of
wetectron/layers/smooth_l1_loss.py:4-16 (reduction=False returns the
element-wise loss, which is how roi_heads/weak_head/loss.py:387-390 calls it)."""
import torch


def smooth_l1_loss(input, target, beta=1.0 / 9, size_average=True, reduction=True):
    d = (input - target).abs()
    loss = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
    if reduction is False:
        return loss
    return loss.mean() if size_average else loss.sum()
