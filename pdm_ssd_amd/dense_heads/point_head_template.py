"""Base of the point-wise heads: layers, target assignment, losses, box decoding.

Host-side restatement (torch) of /root/reference/pcdet/models/dense_heads/point_head_template.py:9-210: same
constructor, `make_fc_layers` layout (Linear(no bias) -> BatchNorm1d -> ReLU per stage, then Linear with bias — hence
the same state_dict keys), same target rules and the same loss arithmetic.  Differences, all in mechanism:
  * foreground / ignore labels of ALL samples come from two batched `points_in_boxes_gpu` calls (the HIP kernel of
    iou3d_nms.hip takes (B, M, 3) points and (B, T, 7) boxes) instead of a python loop with two calls per sample and
    boolean-mask indexing (:82-125); needs the same number of points in every sample, which the PointNet2MSG path
    guarantees (pointnet2_backbone.py:76) — ragged inputs fall back to the per-sample loop;
  * no `.item()` in the loss path (tb_dict holds detached tensors; the reference synchronises three times per step).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..iou3d_nms import iou3d_nms_utils
from ..utils import loss_utils
from ..fused_bn import TrainSequential


def _get(cfg, key, default=None):
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


class PointHeadTemplate(nn.Module):
    def __init__(self, model_cfg, num_class):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.build_losses(_get(self.model_cfg, 'LOSS_CONFIG'))
        self.forward_ret_dict = None

    def build_losses(self, losses_cfg):
        self.add_module('cls_loss_func', loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0))
        reg_loss_type = _get(losses_cfg, 'LOSS_REG', None)
        if reg_loss_type == 'l1':
            self.reg_loss_func = F.l1_loss
        elif reg_loss_type == 'WeightedSmoothL1Loss':
            self.reg_loss_func = loss_utils.WeightedSmoothL1Loss(
                code_weights=_get(_get(losses_cfg, 'LOSS_WEIGHTS'), 'code_weights', None))
        else:   # 'smooth-l1' and the default
            self.reg_loss_func = F.smooth_l1_loss

    @staticmethod
    def make_fc_layers(fc_cfg, input_channels, output_channels):
        layers, c_in = [], input_channels
        for width in fc_cfg:
            layers.extend([nn.Linear(c_in, width, bias=False), nn.BatchNorm1d(width), nn.ReLU()])
            c_in = width
        layers.append(nn.Linear(c_in, output_channels, bias=True))
        return TrainSequential(*layers)   # nn.Sequential whose BatchNorm + ReLU pairs run fused in training (fused_bn.py)

    # ------------------------------------------------------------------ targets

    def assign_stack_targets(self, points, gt_boxes, extend_gt_boxes=None, ret_box_labels=False, ret_part_labels=False,
                             set_ignore_flag=True, use_ball_constraint=False, central_radius=2.0, equal_counts=None):
        """points (N1 + N2 + ..., 4) [bs_idx, x, y, z], gt_boxes (B, M, 8) [box7, class] (zero rows = padding),
        extend_gt_boxes (B, M, 8) -> point_cls_labels (long; 0 background, -1 ignored, else class),
        point_box_labels (.., code_size) of the foreground points' boxes, point_part_labels (.., 3).
        A point inside a box is foreground; inside only the enlarged box: ignored (set_ignore_flag), or foreground only
        within `central_radius` of the box centre (use_ball_constraint)."""
        assert points.dim() == 2 and points.shape[1] == 4, 'points.shape=%s' % str(points.shape)
        assert gt_boxes.dim() == 3 and gt_boxes.shape[2] == 8, 'gt_boxes.shape=%s' % str(gt_boxes.shape)
        assert extend_gt_boxes is None or (extend_gt_boxes.dim() == 3 and extend_gt_boxes.shape[2] == 8), \
            'extend_gt_boxes.shape=%s' % str(extend_gt_boxes.shape)
        assert set_ignore_flag != use_ball_constraint, 'Choose one only!'
        B = gt_boxes.shape[0]
        n_total = points.shape[0]
        # One batched call when every sample has the same point count.  equal_counts=True: the caller has checked that
        # already (PointNet2MSG.forward asserts it; batch_dict['points_per_sample_checked']) — no host sync here.
        # Otherwise test it: with the batch column sorted, the first AND the last row of every n-sized block carrying
        # the block's id is exact (the first row alone lets counts like (3, 5) through).
        n_blk = max(n_total // B, 1)
        if equal_counts is None:
            ids = torch.arange(B, device=points.device)
            equal_counts = n_total % B == 0 and n_total > 0 and bool(
                ((points[::n_blk, 0] == ids) & (points[n_blk - 1::n_blk, 0] == ids)).all())
        if not equal_counts or n_total % B != 0:
            return self._assign_ragged(points, gt_boxes, extend_gt_boxes, ret_box_labels, ret_part_labels, set_ignore_flag,
                                       use_ball_constraint, central_radius)
        n = n_total // B
        xyz = points[:, 1:4].reshape(B, n, 3)
        box_idx = iou3d_nms_utils.points_in_boxes_gpu(xyz, gt_boxes[:, :, 0:7].contiguous()).long()    # (B, n), -1 = none
        fg = box_idx >= 0
        # Everything below is formed for EVERY point and selected with torch.where: indexing with the foreground mask
        # (the reference's form) makes tensors whose size the host must read back, and that read-back stalls the host
        # until the stream has drained — once per mask, in the middle of every training step.
        cls_labels = points.new_zeros((B, n)).long()
        if set_ignore_flag:
            ext_idx = iou3d_nms_utils.points_in_boxes_gpu(xyz, extend_gt_boxes[:, :, 0:7].contiguous())
            cls_labels = torch.where(fg ^ (ext_idx >= 0), cls_labels - 1, cls_labels)
        # the box of every point, (B, n, 8): a row gather over the flattened boxes (torch.gather with an expanded
        # index took 1.2 ms here); background points pick box 0 of their sample, masked out below
        rows = (box_idx.clamp(min=0) + torch.arange(B, device=box_idx.device)[:, None] * gt_boxes.shape[1]).view(-1)
        picked = gt_boxes.reshape(-1, gt_boxes.shape[2]).index_select(0, rows)               # (B n, 8)
        if use_ball_constraint:
            centers = picked[:, 0:3].clone()
            centers[:, 2] += picked[:, 5] / 2
            fg = fg & ((centers.view(B, n, 3) - xyz).norm(dim=-1) < central_radius)
        fg_flat = fg.view(-1)
        classes = picked[:, -1].long()
        cls_flat = torch.where(fg_flat, torch.ones_like(classes) if self.num_class == 1 else classes, cls_labels.view(-1))
        box_labels = part_labels = labels_ok = None
        flat_xyz = xyz.reshape(-1, 3)
        if ret_box_labels:
            # (a background point's row is encoded against a box that is not its own — possibly a zero-sized padding
            #  box, log(0) and all — and then replaced by zeros)
            n_cls = self.box_coder.mean_size.shape[0] if getattr(self.box_coder, 'use_mean_size', False) else None
            codes = self.box_coder.encode_torch(gt_boxes=picked[:, :-1], points=flat_xyz,
                                                gt_classes=classes.clamp(min=1, max=n_cls), check_classes=False)
            box_labels = torch.where(fg_flat[:, None], codes, torch.zeros_like(codes))
            # The reference asserts (with a host synchronisation) that every foreground class addresses the coder's
            # mean-size table; here a class outside it poisons the box targets with NaN, so the step's loss is NaN
            # instead of a silently wrong target — found at the trainer's first look at the loss, no read-back per step.
            ok = self.box_coder.class_range_ok(torch.where(fg_flat, classes, torch.ones_like(classes)))
            box_labels = torch.where(ok, box_labels, torch.full_like(box_labels, float('nan')))
            # (WeightedSmoothL1Loss reads a NaN TARGET as "ignore this element", so the poisoned labels alone would train on
            #  with a zero box loss: the 0-dim flag travels with the targets and get_box_layer_loss turns the loss itself NaN)
            labels_ok = ok
        if ret_part_labels:
            local = flat_xyz - picked[:, 0:3]
            c, s = torch.cos(-picked[:, 6]), torch.sin(-picked[:, 6])
            local = torch.stack((local[:, 0] * c - local[:, 1] * s, local[:, 0] * s + local[:, 1] * c, local[:, 2]), dim=-1)
            part = local / picked[:, 3:6] + 0.5
            part_labels = torch.where(fg_flat[:, None], part, torch.zeros_like(part))
        return {'point_cls_labels': cls_flat, 'point_box_labels': box_labels, 'point_part_labels': part_labels,
                'point_box_labels_ok': labels_ok}

    def _assign_ragged(self, points, gt_boxes, extend_gt_boxes, ret_box_labels, ret_part_labels, set_ignore_flag,
                       use_ball_constraint, central_radius):
        """Samples of different size: one (1, n_k) call per sample, results scattered back by the batch column."""
        B = gt_boxes.shape[0]
        cls_all = points.new_zeros(points.shape[0]).long()
        box_all = gt_boxes.new_zeros((points.shape[0], 8)) if ret_box_labels else None
        part_all = gt_boxes.new_zeros((points.shape[0], 3)) if ret_part_labels else None
        ok_all = None
        for k in range(B):
            sel = (points[:, 0] == k).nonzero().view(-1)
            sub = torch.cat((points.new_zeros((sel.numel(), 1)), points[sel, 1:4]), dim=1)
            t = self.assign_stack_targets(sub, gt_boxes[k:k + 1], None if extend_gt_boxes is None else extend_gt_boxes[k:k + 1],
                                          ret_box_labels, ret_part_labels, set_ignore_flag, use_ball_constraint, central_radius)
            cls_all[sel] = t['point_cls_labels']
            if ret_box_labels:
                box_all[sel] = t['point_box_labels']
                ok_all = t['point_box_labels_ok'] if ok_all is None else ok_all & t['point_box_labels_ok']
            if ret_part_labels:
                part_all[sel] = t['point_part_labels']
        return {'point_cls_labels': cls_all, 'point_box_labels': box_all, 'point_part_labels': part_all,
                'point_box_labels_ok': ok_all}

    # ------------------------------------------------------------------ losses

    def get_cls_layer_loss(self, tb_dict=None):
        """Sigmoid focal loss over all non-ignored points, normalised by the number of foreground points (:141-166)."""
        labels = self.forward_ret_dict['point_cls_labels'].view(-1)
        preds = self.forward_ret_dict['point_cls_preds'].view(-1, self.num_class).float()
        positives = labels > 0
        cls_weights = ((labels == 0) * 1.0 + 1.0 * positives).float()
        pos_normalizer = positives.sum(dim=0).float()
        cls_weights = cls_weights / torch.clamp(pos_normalizer, min=1.0)
        # one-hot over classes 1..C (label 0 = background and the ignored -1 give an all-zero row, as the reference's
        # scatter into C + 1 columns followed by [..., 1:]); a comparison instead of scatter_ (1.2 ms at 524288 points)
        classes = torch.arange(1, self.num_class + 1, device=labels.device)
        one_hot = (labels.unsqueeze(-1) == classes).to(preds.dtype)
        loss = self.cls_loss_func(preds, one_hot, weights=cls_weights).sum()
        loss = loss * _get(_get(self.model_cfg, 'LOSS_CONFIG'), 'LOSS_WEIGHTS')['point_cls_weight']
        tb_dict = {} if tb_dict is None else tb_dict
        tb_dict.update({'point_loss_cls': loss.detach(), 'point_pos_num': pos_normalizer.detach()})
        return loss, tb_dict

    def get_box_layer_loss(self, tb_dict=None):
        """Regression loss over the foreground points only, normalised by their number (:185-206)."""
        pos_mask = self.forward_ret_dict['point_cls_labels'] > 0
        labels = self.forward_ret_dict['point_box_labels']
        preds = self.forward_ret_dict['point_box_preds'].float()
        reg_weights = pos_mask.float()
        reg_weights = reg_weights / torch.clamp(pos_mask.sum().float(), min=1.0)
        if isinstance(self.reg_loss_func, loss_utils.WeightedSmoothL1Loss):
            loss = self.reg_loss_func(preds[None, ...], labels[None, ...], weights=reg_weights[None, ...]).sum()
        else:   # the functional forms take no weights argument
            loss = (self.reg_loss_func(preds, labels, reduction='none') * reg_weights[:, None]).sum()
        loss = loss * _get(_get(self.model_cfg, 'LOSS_CONFIG'), 'LOSS_WEIGHTS')['point_box_weight']
        ok = self.forward_ret_dict.get('point_box_labels_ok')
        if ok is not None:   # a foreground class outside the coder's mean-size table (the reference asserts): NaN loss, NaN gradients
            loss = torch.where(ok, loss, loss.new_full((), float('nan')))
        tb_dict = {} if tb_dict is None else tb_dict
        tb_dict.update({'point_loss_box': loss.detach()})
        return loss, tb_dict

    def generate_predicted_boxes(self, points, point_cls_preds, point_box_preds):
        """points (N, 3), class logits (N, num_class), codes (N, code_size) -> logits, boxes (N, 7) decoded with the
        mean size of each point's arg-max class (:208-222)."""
        _, pred_classes = point_cls_preds.max(dim=-1)
        return point_cls_preds, self.box_coder.decode_torch(point_box_preds, points, pred_classes + 1)

    def forward(self, **kwargs):
        raise NotImplementedError
