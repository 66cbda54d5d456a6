"""Point-anchored residual box coder of the point head.

Behaviour of /root/reference/pcdet/utils/box_coder_utils.py:144-222 (PointResidualCoder): a box is coded relative to
the POINT that votes for it — centre offset divided by the class's mean-size diagonal (xy) / height (z), log size
ratios, and the heading as (cos, sin).  The reference moves `mean_size` to the GPU in the constructor; here it
follows the device of the tensors it is used with.
"""
import numpy as np
import torch


class PointResidualCoder(object):
    def __init__(self, code_size=8, use_mean_size=True, **kwargs):
        super().__init__()
        self.code_size = code_size
        self.use_mean_size = use_mean_size
        if self.use_mean_size:
            self.mean_size = torch.from_numpy(np.array(kwargs['mean_size'])).float()
            assert self.mean_size.min() > 0

    def _anchor(self, classes, like, check=False):
        if self.mean_size.device != like.device:
            self.mean_size = self.mean_size.to(like.device)
        if check:   # (a host synchronisation: only where the classes come from data, i.e. target encoding)
            assert classes.max() <= self.mean_size.shape[0]
        size = self.mean_size[classes - 1]
        dxa, dya, dza = torch.split(size, 1, dim=-1)
        return dxa, dya, dza, torch.sqrt(dxa ** 2 + dya ** 2)

    def encode_torch(self, gt_boxes, points, gt_classes=None, check_classes=True):
        """gt_boxes (N, 7 + C), points (N, 3), gt_classes (N) in [1, num_classes] -> (N, 8 + C).
        As in the reference (:164) the sizes of `gt_boxes` are clamped IN PLACE to >= 1e-5.
        check_classes: the reference's range assert on the classes (:166), a device->host synchronisation; the point
        head's per-step target encoding passes False and checks the range on the device instead (class_range_ok)."""
        gt_boxes[:, 3:6] = torch.clamp_min(gt_boxes[:, 3:6], min=1e-5)
        xg, yg, zg, dxg, dyg, dzg, rg, *cgs = torch.split(gt_boxes, 1, dim=-1)
        xa, ya, za = torch.split(points, 1, dim=-1)
        if self.use_mean_size:
            dxa, dya, dza, diagonal = self._anchor(gt_classes, gt_boxes, check=check_classes)
            xt, yt, zt = (xg - xa) / diagonal, (yg - ya) / diagonal, (zg - za) / dza
            dxt, dyt, dzt = torch.log(dxg / dxa), torch.log(dyg / dya), torch.log(dzg / dza)
        else:
            xt, yt, zt = xg - xa, yg - ya, zg - za
            dxt, dyt, dzt = torch.log(dxg), torch.log(dyg), torch.log(dzg)
        return torch.cat([xt, yt, zt, dxt, dyt, dzt, torch.cos(rg), torch.sin(rg), *cgs], dim=-1)

    def class_range_ok(self, classes):
        """0-dim bool tensor on the device: every class id addresses a row of the mean-size table (no synchronisation)."""
        if not self.use_mean_size:
            return torch.ones((), dtype=torch.bool, device=classes.device)
        return classes.max() <= self.mean_size.shape[0]

    def decode_torch(self, box_encodings, points, pred_classes=None):
        """box_encodings (N, 8 + C) [x, y, z, dx, dy, dz, cos, sin, ...], points (N, 3) -> boxes (N, 7 + C).
        pred_classes in [1, num_classes] (an arg-max + 1 in the head: the reference's range assert, a device->host
        synchronisation per call, is kept for the encoder only, so decoding can be captured in a hipGraph)."""
        xt, yt, zt, dxt, dyt, dzt, cost, sint, *cts = torch.split(box_encodings, 1, dim=-1)
        xa, ya, za = torch.split(points, 1, dim=-1)
        if self.use_mean_size:
            dxa, dya, dza, diagonal = self._anchor(pred_classes, box_encodings)
            xg, yg, zg = xt * diagonal + xa, yt * diagonal + ya, zt * dza + za
            dxg, dyg, dzg = torch.exp(dxt) * dxa, torch.exp(dyt) * dya, torch.exp(dzt) * dza
        else:
            xg, yg, zg = xt + xa, yt + ya, zt + za
            dxg, dyg, dzg = torch.split(torch.exp(box_encodings[..., 3:6]), 1, dim=-1)
        rg = torch.atan2(sint, cost)
        return torch.cat([xg, yg, zg, dxg, dyg, dzg, rg, *cts], dim=-1)
