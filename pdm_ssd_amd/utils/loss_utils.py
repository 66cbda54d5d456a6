"""Losses of the hybrid head.

SigmoidFocalClassificationLoss, WeightedSmoothL1Loss: behaviour of /root/reference/pcdet/utils/loss_utils.py:10-74,
76-141 (point head).  FocalLossCenterNet: :266-345 (heat-map head; the CornerNet penalty-reduced focal loss).
Code weights follow the device of the input instead of being moved to the GPU at construction.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class SigmoidFocalClassificationLoss(nn.Module):
    """Focal loss on logits: balance(t) * miss(t, p)^gamma * BCE(x, t) per element, where p = sigmoid(x),
    miss = the probability mass on the wrong side, balance = alpha for positives and 1 - alpha for negatives."""

    def __init__(self, gamma: float = 2.0, alpha: float = 0.25):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    @staticmethod
    def sigmoid_cross_entropy_with_logits(input, target):
        """Binary cross entropy of logits in its overflow-free form (kept under the reference's name, :26-43)."""
        return F.binary_cross_entropy_with_logits(input, target, reduction='none')

    def forward(self, input, target, weights):
        """input / target (B, #anchors, #classes) logits / one-hot, weights (B, #anchors) -> unreduced loss."""
        if target.dtype != input.dtype:   # mixed dtypes promote, as the reference's arithmetic form does (lerp would raise)
            dt = torch.promote_types(input.dtype, target.dtype)
            input, target = input.to(dt), target.to(dt)
        p = torch.sigmoid(input)
        miss = torch.lerp(p, 1.0 - p, target)                   # t (1 - p) + (1 - t) p
        balance = (1.0 - self.alpha) + target * (2.0 * self.alpha - 1.0)
        per_elem = balance * miss.pow(self.gamma) * self.sigmoid_cross_entropy_with_logits(input, target)
        if weights.dim() + 1 == per_elem.dim():                  # one weight per anchor: broadcast over the classes
            weights = weights[..., None]
        assert weights.dim() == per_elem.dim()
        return per_elem * weights


class WeightedSmoothL1Loss(nn.Module):
    """Huber-style regression loss with the knee at beta (default 1/9): quadratic inside, linear outside; residuals are
    scaled per code before the loss and per anchor after it; a NaN target switches its element off."""

    def __init__(self, beta: float = 1.0 / 9.0, code_weights: list = None):
        super().__init__()
        self.beta = beta
        self.code_weights = None if code_weights is None else torch.from_numpy(np.array(code_weights, dtype=np.float32))

    @staticmethod
    def smooth_l1_loss(diff, beta):
        mag = diff.abs()
        if beta < 1e-5:          # degenerate knee: plain L1
            return mag
        return torch.where(mag < beta, mag * mag * (0.5 / beta), mag - 0.5 * beta)

    def forward(self, input, target, weights=None):
        """input / target (B, #anchors, #codes), weights (B, #anchors) -> (B, #anchors, #codes) unreduced."""
        residual = torch.where(torch.isnan(target), torch.zeros_like(input), input - target)
        if self.code_weights is not None:
            self.code_weights = self.code_weights.to(residual.device)
            residual = residual * self.code_weights
        out = self.smooth_l1_loss(residual, self.beta)
        if weights is not None:
            assert weights.shape[:2] == out.shape[:2]
            out = out * weights[..., None]
        return out


def neg_loss_cornernet(pred, gt, mask=None):
    """Penalty-reduced pixel focal loss of the heat-map head.  pred / gt (B, C, H, W) in (0, 1) / [0, 1]: peaks
    (gt == 1) contribute (1-p)^2 log p, every other cell (1-gt)^4 p^2 log(1-p); the sum is negated and divided by the
    number of peaks (by 1 when there is none, which leaves the negative term alone)."""
    peak = gt == 1
    zero = torch.zeros_like(pred)
    at_peaks = torch.where(peak, (1 - pred).square() * pred.log(), zero)
    elsewhere = torch.where(peak, zero, (1 - gt).pow(4) * pred.square() * (1 - pred).log())
    peaks = peak.to(pred.dtype)
    if mask is not None:
        keep = mask[:, None].to(pred.dtype)
        at_peaks, elsewhere, peaks = at_peaks * keep, elsewhere * keep, peaks * keep
    # (the reference branches on the host, `if num_pos == 0`: a device->host sync per step; same value here)
    return -(at_peaks.sum() + elsewhere.sum()) / peaks.sum().clamp(min=1.0)


class FocalLossCenterNet(nn.Module):
    def forward(self, out, target, mask=None):
        return neg_loss_cornernet(out, target, mask=mask)
