"""Heat-map head in training: gaussian targets and the penalty-reduced focal loss with its gradient as two operators
(csrc/heatmap_loss.hip) instead of ~75 elementwise torch kernels per step.

The torch formulation stays in dense_heads/pdm_heatmap_head.py::assign_targets, utils/centernet_utils.py and
utils/loss_utils.py::neg_loss_cornernet (behaviour of /root/reference/pcdet/models/dense_heads/center_head.py:100-160, :232,
/root/reference/pcdet/utils/loss_utils.py:266-304); the forward pass here also leaves d S / d logit, and backward() only scales it.
"""
import torch
from torch.autograd import Function

from . import _native


@torch.no_grad()
def heatmap_targets(gt_boxes, num_class, H, W, x0, y0, vx, vy, stride, min_overlap, min_radius, max_radius):
    """gt_boxes (B, M, 8) fp32 on the GPU -> (B, num_class, H, W) fp32 targets (pdm_heatmap_targets)."""
    gt = gt_boxes.detach().float().contiguous()
    B, M, _ = gt.shape
    hm = torch.empty((B, num_class, H, W), dtype=torch.float32, device=gt.device)
    _native.call("pdm_heatmap_targets", torch.cuda.current_stream(gt.device).cuda_stream, B, M, num_class, H, W, gt.data_ptr(),
                 float(x0), float(y0), float(vx), float(vy), float(stride), float(min_overlap), int(min_radius), int(max_radius), hm.data_ptr())
    return hm


class _HeatmapFocalLoss(Function):
    @staticmethod
    def forward(ctx, logits, heatmap, weight):
        """logits (B, C, H, W) fp32 or bf16, any strides; heatmap (B, C, H, W) fp32 contiguous -> 0-dim fp32 loss."""
        assert logits.dim() == 4 and logits.dtype in (torch.float32, torch.bfloat16)
        assert heatmap.shape == logits.shape and heatmap.dtype == torch.float32 and heatmap.is_contiguous()
        B, C, H, W = logits.shape
        dev = logits.device
        l = _native.lib()
        nbytes = l.pdm_heatmap_focal_loss_workspace_bytes(logits.numel())
        ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)
        dl = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        sb, sc, sh, sw = logits.stride()
        _native.call("pdm_heatmap_focal_loss", torch.cuda.current_stream(dev).cuda_stream, B, C, H, W, logits.data_ptr(),
                     1 if logits.dtype == torch.bfloat16 else 0, sb, sc, sh, sw, heatmap.data_ptr(), float(weight), dl.data_ptr(),
                     out.data_ptr(), ws.data_ptr(), nbytes)
        ctx.save_for_backward(dl, out)
        ctx.in_dtype = logits.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        dl, out = ctx.saved_tensors
        return (dl * (g.float() * out[1])).to(ctx.in_dtype), None, None


def heatmap_focal_loss(logits, heatmap, weight=1.0):
    return _HeatmapFocalLoss.apply(logits, heatmap, float(weight))
