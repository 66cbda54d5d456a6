"""Point-wise box head: the point half of PDM-SSD's hybrid head.

Host-side restatement of /root/reference/pcdet/models/dense_heads/point_head_box.py:7-115 (PointHeadBox): same
constructor (num_class, input_channels, model_cfg, predict_boxes_when_training), config keys (CLS_FC, REG_FC,
TARGET_CONFIG.{GT_EXTRA_WIDTH, BOX_CODER, BOX_CODER_CONFIG}, LOSS_CONFIG), batch_dict keys in and out, state_dict keys
(`cls_layers.*`, `box_layers.*`).  In eval mode without autograd the two MLPs run as per-row fp32 MFMA kernels
(pdm_rows_mlp_fused, BatchNorm folded) on the point-major features the backbone hands over; training uses the torch
layers.
"""
import torch

from .. import fused
from ..utils import box_coder_utils, box_utils
from .point_head_template import PointHeadTemplate, _get


class _AsConv:
    """nn.Linear seen as the 1x1 convolution fused.PackedMLP folds and packs."""

    def __init__(self, linear):
        self.weight, self.bias = linear.weight, linear.bias
        self.out_channels, self.in_channels = linear.out_features, linear.in_features


def _fc_layers(seq):
    """make_fc_layers' Sequential -> [(linear, bn | None)] (the last Linear has no BatchNorm / ReLU behind it)."""
    mods, out, i = list(seq), [], 0
    while i < len(mods):
        lin = mods[i]
        assert isinstance(lin, torch.nn.Linear)
        if i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.BatchNorm1d):
            out.append((_AsConv(lin), mods[i + 1]))
            i += 3
        else:
            out.append((_AsConv(lin), None))
            i += 1
    return out


class PointHeadBox(PointHeadTemplate):
    def __init__(self, num_class, input_channels, model_cfg, predict_boxes_when_training=False, **kwargs):
        super().__init__(model_cfg=model_cfg, num_class=num_class)
        self.predict_boxes_when_training = predict_boxes_when_training
        self.cls_layers = self.make_fc_layers(fc_cfg=_get(self.model_cfg, 'CLS_FC'), input_channels=input_channels,
                                              output_channels=num_class)
        target_cfg = _get(self.model_cfg, 'TARGET_CONFIG')
        self.box_coder = getattr(box_coder_utils, _get(target_cfg, 'BOX_CODER'))(**_get(target_cfg, 'BOX_CODER_CONFIG'))
        self.box_layers = self.make_fc_layers(fc_cfg=_get(self.model_cfg, 'REG_FC'), input_channels=input_channels,
                                              output_channels=self.box_coder.code_size)

    def assign_targets(self, input_dict):
        """point_coords (N1 + N2 + ..., 4) [bs_idx, x, y, z], gt_boxes (B, M, 8) -> point_cls_labels (0 background,
        -1 ignored: inside the box enlarged by GT_EXTRA_WIDTH but not the box), point_box_labels (ref :31-59)."""
        point_coords = input_dict['point_coords']
        gt_boxes = input_dict['gt_boxes']
        assert gt_boxes.dim() == 3, 'gt_boxes.shape=%s' % str(gt_boxes.shape)
        assert point_coords.dim() == 2, 'points.shape=%s' % str(point_coords.shape)
        batch_size = gt_boxes.shape[0]
        extend_gt_boxes = box_utils.enlarge_box3d(
            gt_boxes.view(-1, gt_boxes.shape[-1]), extra_width=_get(_get(self.model_cfg, 'TARGET_CONFIG'), 'GT_EXTRA_WIDTH')
        ).view(batch_size, -1, gt_boxes.shape[-1])
        return self.assign_stack_targets(points=point_coords, gt_boxes=gt_boxes, extend_gt_boxes=extend_gt_boxes,
                                         set_ignore_flag=True, use_ball_constraint=False,
                                         ret_part_labels=False, ret_box_labels=True,
                                         equal_counts=True if input_dict.get('points_per_sample_checked', False) else None)

    def _fused_loss_inputs(self, batch_dict, cls_preds, box_preds):
        """Training on the GPU with the standard configuration (mean-size PointResidualCoder, WeightedSmoothL1Loss, equal point
        counts): the inputs of pdm_point_head_loss — targets, both losses and their gradients as one operator
        (pdm_ssd_amd/head_loss.py) instead of ~100 elementwise kernels.  None when the configuration differs (the torch
        formulation below runs then; `use_fused_loss = False` forces it)."""
        from ..utils import loss_utils
        from ..iou3d_nms import iou3d_nms_utils
        coords, gt_boxes = batch_dict['point_coords'], batch_dict.get('gt_boxes')
        coder = self.box_coder
        if (not getattr(self, 'use_fused_loss', True) or not cls_preds.is_cuda or gt_boxes is None or gt_boxes.dim() != 3
                or gt_boxes.shape[2] != 8 or gt_boxes.shape[1] < 1 or cls_preds.dtype != box_preds.dtype
                or cls_preds.dtype not in (torch.float32, torch.bfloat16) or cls_preds.dim() != 2 or cls_preds.stride(1) != 1
                or box_preds.stride(1) != 1 or not isinstance(coder, box_coder_utils.PointResidualCoder) or not coder.use_mean_size
                or coder.code_size != 8 or not isinstance(self.reg_loss_func, loss_utils.WeightedSmoothL1Loss)
                or not batch_dict.get('points_per_sample_checked', False) or coords.dtype != torch.float32 or coords.stride(1) != 1
                or coords.shape[0] % gt_boxes.shape[0] != 0 or coords.shape[0] == 0):
            return None
        B = gt_boxes.shape[0]
        n = coords.shape[0] // B
        gt_boxes = gt_boxes.float().contiguous()
        extend = box_utils.enlarge_box3d(gt_boxes.view(-1, 8), extra_width=_get(_get(self.model_cfg, 'TARGET_CONFIG'), 'GT_EXTRA_WIDTH')
                                         ).view(B, -1, 8)
        xyz = coords[:, 1:4].reshape(B, n, 3)
        box_idx = iou3d_nms_utils.points_in_boxes_gpu(xyz, gt_boxes[:, :, 0:7].contiguous()).view(-1)
        ext_idx = iou3d_nms_utils.points_in_boxes_gpu(xyz, extend[:, :, 0:7].contiguous()).view(-1)
        if coder.mean_size.device != cls_preds.device:
            coder.mean_size = coder.mean_size.to(cls_preds.device)
        return {'xyz_rows': coords[:, 1:4], 'box_idx': box_idx, 'ext_idx': ext_idx, 'gt_boxes': gt_boxes, 'n': n}

    def _fused_loss(self, tb_dict):
        from .. import head_loss
        fr = self.forward_ret_dict
        f = fr['fused_loss_inputs']
        w = _get(_get(self.model_cfg, 'LOSS_CONFIG'), 'LOSS_WEIGHTS')
        cw = self.reg_loss_func.code_weights
        cw = [1.0] * 8 if cw is None else [float(v) for v in cw]
        loss_cls, loss_box, pos, labels = head_loss.point_head_loss(
            fr['point_cls_preds'], fr['point_box_preds'], f['xyz_rows'], f['box_idx'], f['ext_idx'], f['gt_boxes'],
            self.box_coder.mean_size.float().contiguous(), f['n'], cw, self.reg_loss_func.beta, self.cls_loss_func.alpha,
            self.cls_loss_func.gamma, w['point_cls_weight'], w['point_box_weight'])
        fr['point_cls_labels'] = labels
        tb_dict.update({'point_loss_cls': loss_cls.detach(), 'point_pos_num': pos, 'point_loss_box': loss_box.detach()})
        return loss_cls + loss_box, tb_dict

    def get_loss(self, tb_dict=None):
        tb_dict = {} if tb_dict is None else tb_dict
        if self.forward_ret_dict.get('fused_loss_inputs') is not None:
            return self._fused_loss(tb_dict)
        point_loss_cls, tb_dict_1 = self.get_cls_layer_loss()
        point_loss_box, tb_dict_2 = self.get_box_layer_loss()
        tb_dict.update(tb_dict_1)
        tb_dict.update(tb_dict_2)
        return point_loss_cls + point_loss_box, tb_dict

    def _layers(self, point_features, deferred=None):
        """The two MLPs.  Inference: per-row fused kernels (BatchNorm folded) when the features are fp32 rows on the
        GPU; otherwise the torch layers.  deferred: the backbone's last FP module not yet run (fused.DeferredFP), whose
        output point_features is: run with the two stacks in one launch when the shapes fit, else filled first."""
        if deferred is not None and not self._fp_fusable(point_features, deferred):
            deferred.materialize()
            deferred = None
        infer = (not self.training and not torch.is_grad_enabled() and point_features.is_cuda
                 and point_features.dtype == torch.float32 and point_features.dim() == 2
                 and getattr(self, 'use_fused', True))
        if infer:
            pc = fused.cached_layers(self, 'cls', self.cls_layers, lambda: _fc_layers(self.cls_layers), point_features.device)
            pb = fused.cached_layers(self, 'box', self.box_layers, lambda: _fc_layers(self.box_layers), point_features.device)
            if pc is not None and pb is not None:
                rows = point_features.contiguous().unsqueeze(0)
                # (the kernel writes 16-byte groups: row strides padded to a multiple of 4 floats)
                ncls, nbox = self.num_class, self.box_coder.code_size
                cls = torch.empty((1, rows.shape[1], (ncls + 3) // 4 * 4), dtype=torch.float32, device=rows.device)
                box = torch.empty((1, rows.shape[1], (nbox + 3) // 4 * 4), dtype=torch.float32, device=rows.device)
                if getattr(self, 'use_x3', False) and list(pc.dims) == [128, 256, 256, 16] == list(pb.dims):
                    # OPT-IN: fp32 emulated on the bf16 matrix pipe (three bf16 pieces per operand, six partial products,
                    # fp32 accumulation: csrc/rows_chain_x3.hip); agrees with the fp32-MFMA kernels to ~1e-6 of the output scale
                    xc = fused.cached_layers_x3(self, 'cls_x3', self.cls_layers, lambda: _fc_layers(self.cls_layers), point_features.device)
                    xb = fused.cached_layers_x3(self, 'box_x3', self.box_layers, lambda: _fc_layers(self.box_layers), point_features.device)
                    fused.rows_forward_x3(xc, rows, cls, relu_last=False)
                    fused.rows_forward_x3(xb, rows, box, relu_last=False)
                elif deferred is not None:
                    # FP module + both stacks in one launch: the rows are written once and not read back
                    fused.fp_head_forward(deferred, pc, pb, cls, box, relu_last=False)
                elif list(pc.dims) == list(pb.dims) and getattr(self, 'use_pair', True):
                    # one launch for both stacks (the rows are read once); bit-identical to the two calls below
                    fused.rows_forward_pair(pc, pb, rows, cls, box, relu_last=False)
                else:
                    fused.rows_forward(pc, rows, cls, relu_last=False)
                    fused.rows_forward(pb, rows, box, relu_last=False)
                return cls[0, :, :ncls], box[0, :, :nbox]
        return self.cls_layers(point_features), self.box_layers(point_features)

    def _fp_fusable(self, point_features, deferred):
        """May the deferred FP module run inside the point head's launch (pdm_fp_head_fused)?"""
        if (self.training or torch.is_grad_enabled() or not point_features.is_cuda or point_features.dtype != torch.float32
                or point_features.dim() != 2 or not getattr(self, 'use_fused', True) or not getattr(self, 'use_pair', True)
                or getattr(self, 'use_x3', False) or not getattr(self, 'use_fp_fusion', False)
                or point_features.data_ptr() != deferred.out_pm.data_ptr()):
            return False
        pc = fused.cached_layers(self, 'cls', self.cls_layers, lambda: _fc_layers(self.cls_layers), point_features.device)
        pb = fused.cached_layers(self, 'box', self.box_layers, lambda: _fc_layers(self.box_layers), point_features.device)
        return pc is not None and pb is not None and deferred.fits_head(pc, pb)

    def wants_deferred_fp(self):
        """Should the backbone leave its last FP module to this head (batch_dict['defer_last_fp'])?  OPT-IN (use_fp_fusion):
        the one-launch form is bit-identical to the separate launches and NOT faster at the bench shape — 2.14 ms against
        2.10 for the last FP module + both stacks + decode (tools/diag/fp_head_rate.py): inside a workgroup the module's
        gathers, staging and stores still run ahead of its MFMAs, tile by tile, as they do in its own launch."""
        return (not self.training and not torch.is_grad_enabled() and getattr(self, 'use_fused', True) and getattr(self, 'use_pair', True)
                and not getattr(self, 'use_x3', False) and getattr(self, 'use_fp_fusion', False)
                and not _get(self.model_cfg, 'USE_POINT_FEATURES_BEFORE_FUSION', False))

    def _decode_fused(self, batch_dict, cls, box):
        """Eval mode on the GPU: scores and decoded boxes of all points in one HIP pass (pdm_point_head_decode) instead
        of ~25 elementwise torch kernels.  False when the configuration is not the fp32 mean-size coder path."""
        from .. import _native
        coords = batch_dict['point_coords']
        coder = self.box_coder
        if (self.training or torch.is_grad_enabled() or not cls.is_cuda or cls.dtype != torch.float32 or box.dtype != torch.float32
                or not isinstance(coder, box_coder_utils.PointResidualCoder) or not coder.use_mean_size or coder.code_size != 8
                or coords.dtype != torch.float32 or coords.stride(1) != 1 or cls.stride(1) != 1 or box.stride(1) != 1
                or box.stride(0) % 4 or box.data_ptr() % 16 or not getattr(self, 'use_fused', True)):
            return False
        if coder.mean_size.device != cls.device:
            coder.mean_size = coder.mean_size.to(cls.device)
        n = cls.shape[0]
        boxes = torch.empty((n, 7), dtype=torch.float32, device=cls.device)
        scores = torch.empty((n,), dtype=torch.float32, device=cls.device)
        pts = coords[:, 1:4]
        _native.call("pdm_point_head_decode", torch.cuda.current_stream(cls.device).cuda_stream, n, self.num_class, cls.data_ptr(),
                     cls.stride(0), box.data_ptr(), box.stride(0), pts.data_ptr(), coords.stride(0),
                     coder.mean_size.contiguous().data_ptr(), boxes.data_ptr(), scores.data_ptr())
        batch_dict['point_cls_scores'] = scores
        batch_dict['batch_cls_preds'] = cls
        batch_dict['batch_box_preds'] = boxes
        batch_dict['batch_index'] = coords[:, 0]
        batch_dict['cls_preds_normalized'] = False
        return True

    def forward(self, batch_dict):
        """point_features (N1 + N2 + ..., C), point_coords (.., 4) [, gt_boxes (B, M, 8)] -> point_cls_scores and, in
        eval mode (or predict_boxes_when_training), batch_cls_preds / batch_box_preds / batch_index (ref :71-115)."""
        if _get(self.model_cfg, 'USE_POINT_FEATURES_BEFORE_FUSION', False):
            point_features = batch_dict['point_features_before_fusion']
        else:
            point_features = batch_dict['point_features']
        deferred = batch_dict.pop('point_features_deferred', None)
        if deferred is not None and point_features is not batch_dict.get('point_features'):
            deferred.materialize()      # this head reads other rows: the module's output is still owed to the detector
            deferred = None
        point_cls_preds, point_box_preds = self._layers(point_features, deferred)
        if self._decode_fused(batch_dict, point_cls_preds, point_box_preds):
            self.forward_ret_dict = {'point_cls_preds': point_cls_preds, 'point_box_preds': point_box_preds}
            return batch_dict
        point_cls_preds_max, _ = point_cls_preds.max(dim=-1)
        batch_dict['point_cls_scores'] = torch.sigmoid(point_cls_preds_max)
        ret_dict = {'point_cls_preds': point_cls_preds, 'point_box_preds': point_box_preds}
        fused_in = self._fused_loss_inputs(batch_dict, point_cls_preds, point_box_preds) if self.training else None
        if fused_in is not None:
            ret_dict['fused_loss_inputs'] = fused_in      # labels follow from get_loss() (forward_ret_dict['point_cls_labels'])
        elif self.training:
            targets_dict = self.assign_targets(batch_dict)
            ret_dict['point_cls_labels'] = targets_dict['point_cls_labels']
            ret_dict['point_box_labels'] = targets_dict['point_box_labels']
            ret_dict['point_box_labels_ok'] = targets_dict.get('point_box_labels_ok')
        if not self.training or self.predict_boxes_when_training:
            point_cls_preds, point_box_preds = self.generate_predicted_boxes(
                points=batch_dict['point_coords'][:, 1:4], point_cls_preds=point_cls_preds, point_box_preds=point_box_preds)
            batch_dict['batch_cls_preds'] = point_cls_preds
            batch_dict['batch_box_preds'] = point_box_preds
            batch_dict['batch_index'] = batch_dict['point_coords'][:, 0]
            batch_dict['cls_preds_normalized'] = False
        self.forward_ret_dict = ret_dict
        return batch_dict
