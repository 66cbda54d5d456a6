"""Losses of the hybrid head.

SigmoidFocalClassificationLoss, WeightedSmoothL1Loss: behaviour of /root/reference/pcdet/utils/loss_utils.py:10-74,
76-141 (point head).  FocalLossCenterNet: :266-345 (heat-map head; the CornerNet penalty-reduced focal loss).
Code weights follow the device of the input instead of being moved to the GPU at construction.
"""
import numpy as np
import torch
import torch.nn as nn


class SigmoidFocalClassificationLoss(nn.Module):
    def __init__(self, gamma: float = 2.0, alpha: float = 0.25):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    @staticmethod
    def sigmoid_cross_entropy_with_logits(input, target):
        """max(x, 0) - x z + log(1 + exp(-|x|)) — the numerically stable form (:26-43)."""
        return torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))

    def forward(self, input, target, weights):
        """input / target (B, #anchors, #classes) logits / one-hot, weights (B, #anchors) -> unreduced loss."""
        pred_sigmoid = torch.sigmoid(input)
        alpha_weight = target * self.alpha + (1 - target) * (1 - self.alpha)
        pt = target * (1.0 - pred_sigmoid) + (1.0 - target) * pred_sigmoid
        focal_weight = alpha_weight * torch.pow(pt, self.gamma)
        loss = focal_weight * self.sigmoid_cross_entropy_with_logits(input, target)
        if weights.dim() == 2 or (weights.dim() == 1 and target.dim() == 2):
            weights = weights.unsqueeze(-1)
        assert weights.dim() == loss.dim()
        return loss * weights


class WeightedSmoothL1Loss(nn.Module):
    """smooth-L1 with change point beta = 1/9, code-wise weights, anchor-wise weights; NaN targets are ignored."""

    def __init__(self, beta: float = 1.0 / 9.0, code_weights: list = None):
        super().__init__()
        self.beta = beta
        self.code_weights = None if code_weights is None else torch.from_numpy(np.array(code_weights, dtype=np.float32))

    @staticmethod
    def smooth_l1_loss(diff, beta):
        if beta < 1e-5:
            return torch.abs(diff)
        n = torch.abs(diff)
        return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)

    def forward(self, input, target, weights=None):
        """input / target (B, #anchors, #codes), weights (B, #anchors) -> (B, #anchors, #codes) unreduced."""
        target = torch.where(torch.isnan(target), input, target)
        diff = input - target
        if self.code_weights is not None:
            if self.code_weights.device != diff.device:
                self.code_weights = self.code_weights.to(diff.device)
            diff = diff * self.code_weights.view(1, 1, -1)
        loss = self.smooth_l1_loss(diff, self.beta)
        if weights is not None:
            assert weights.shape[0] == loss.shape[0] and weights.shape[1] == loss.shape[1]
            loss = loss * weights.unsqueeze(-1)
        return loss


def neg_loss_cornernet(pred, gt, mask=None):
    """pred / gt (B, C, H, W) in (0, 1) / [0, 1]: -(sum over gt == 1 of log(p)(1-p)^2 + sum over gt < 1 of
    log(1-p) p^2 (1-gt)^4) / #positives (the negative term alone when there is no positive)."""
    pos_inds = gt.eq(1).float()
    neg_inds = gt.lt(1).float()
    neg_weights = torch.pow(1 - gt, 4)
    pos_loss = torch.log(pred) * torch.pow(1 - pred, 2) * pos_inds
    neg_loss = torch.log(1 - pred) * torch.pow(pred, 2) * neg_weights * neg_inds
    if mask is not None:
        mask = mask[:, None, :, :].float()
        pos_loss, neg_loss = pos_loss * mask, neg_loss * mask
        num_pos = (pos_inds * mask).sum()
    else:
        num_pos = pos_inds.sum()
    pos_loss, neg_loss = pos_loss.sum(), neg_loss.sum()
    # (the reference branches on the host, `if num_pos == 0`: a device->host sync per step; same value here)
    return -(pos_loss + neg_loss) / torch.clamp(num_pos, min=1.0)


class FocalLossCenterNet(nn.Module):
    def forward(self, out, target, mask=None):
        return neg_loss_cornernet(out, target, mask=mask)
