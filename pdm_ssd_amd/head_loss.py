"""Point head: target assignment, focal + smooth-L1 losses and their gradients as one operator (csrc/head_loss.hip).

The torch formulation in dense_heads/point_head_template.py (behaviour of
/root/reference/pcdet/models/dense_heads/point_head_template.py:51-206) is ~100 elementwise kernels per step over the
(B n, 3) / (B n, 8) prediction tensors; here the forward pass also leaves d L / d pred, and backward() only scales
them by the incoming gradients of the two loss scalars.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _native


class _PointHeadLoss(Function):
    @staticmethod
    def forward(ctx, cls_preds, box_preds, xyz_rows, box_idx, ext_idx, gt_boxes, mean_size, spec):
        """cls_preds (N, C), box_preds (N, 8) fp32 or bf16 (same dtype, unit inner stride); xyz_rows (N, >= 3) fp32 view whose
        columns 0..2 are x, y, z; box_idx / ext_idx (N) int32; gt_boxes (B, M, 8) fp32; mean_size (n_mean, 3) fp32;
        spec = (n_per_sample, code_weights[8], beta, alpha, gamma, cls_weight, box_weight).
        Returns (loss_cls, loss_box, n_positives, labels): three 0-dim fp32 tensors and (N) int64 labels."""
        n_per_sample, code_w, beta, alpha, gamma, cls_w, box_w = spec
        N, C = cls_preds.shape
        assert cls_preds.dtype == box_preds.dtype and cls_preds.dtype in (torch.float32, torch.bfloat16)
        assert cls_preds.stride(1) == 1 and box_preds.stride(1) == 1 and box_preds.shape == (N, 8)
        assert xyz_rows.dtype == torch.float32 and xyz_rows.stride(1) == 1 and gt_boxes.is_contiguous() and gt_boxes.shape[2] == 8
        assert box_idx.dtype == torch.int32 and ext_idx.dtype == torch.int32 and box_idx.is_contiguous() and ext_idx.is_contiguous()
        dev = cls_preds.device
        l = _native.lib()
        nbytes = l.pdm_point_head_loss_workspace_bytes(N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        labels = torch.empty(N, dtype=torch.int64, device=dev)
        dcls = torch.empty((N, C), dtype=cls_preds.dtype, device=dev)
        dbox = torch.empty((N, 8), dtype=box_preds.dtype, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        cw = (ctypes.c_float * 8)(*[float(v) for v in code_w])
        _native.call("pdm_point_head_loss", torch.cuda.current_stream(dev).cuda_stream, N, n_per_sample, gt_boxes.shape[1], C,
                     mean_size.shape[0], 1 if cls_preds.dtype == torch.bfloat16 else 0, cls_preds.data_ptr(), cls_preds.stride(0),
                     box_preds.data_ptr(), box_preds.stride(0), xyz_rows.data_ptr(), xyz_rows.stride(0), box_idx.data_ptr(),
                     ext_idx.data_ptr(), gt_boxes.data_ptr(), mean_size.data_ptr(), ctypes.cast(cw, ctypes.c_void_p), beta, alpha, gamma,
                     cls_w, box_w, labels.data_ptr(), dcls.data_ptr(), dbox.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes)
        ctx.save_for_backward(dcls, dbox)
        ctx.mark_non_differentiable(labels)
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2].detach(), labels

    @staticmethod
    def backward(ctx, g_cls, g_box, _g_pos, _g_labels):
        dcls, dbox = ctx.saved_tensors
        gc = None if g_cls is None else dcls * g_cls.to(dcls.dtype)
        gb = None if g_box is None else dbox * g_box.to(dbox.dtype)
        return gc, gb, None, None, None, None, None, None


def point_head_loss(cls_preds, box_preds, xyz_rows, box_idx, ext_idx, gt_boxes, mean_size, n_per_sample, code_weights,
                    beta, alpha, gamma, cls_weight, box_weight):
    return _PointHeadLoss.apply(cls_preds, box_preds, xyz_rows, box_idx, ext_idx, gt_boxes, mean_size,
                                (int(n_per_sample), list(code_weights), float(beta), float(alpha), float(gamma), float(cls_weight),
                                 float(box_weight)))
