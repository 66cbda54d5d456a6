"""BEV heat-map head: the dense half of PDM-SSD's hybrid head ("Context Learning -> Heatmap Predict" in the
reference's docs/workflow.svg; README.md:12: "predict the scene heatmap ... to supplement the voting point set").

No source for it exists in the snapshot, so this module is build-defined on the reference's own
pieces for the same job: the detector's DENSE_HEAD slot and constructor keywords
(/root/reference/pcdet/models/detectors/detector3d_template.py:113-139), CenterPoint-style gaussian targets
(dense_heads/center_head.py:106-160 + model_utils/centernet_utils.py:9-70) and the penalty-reduced focal loss
(utils/loss_utils.py:266-345, weight `cls_weight` as center_head.py:238-242).  Plain torch convolutions: the BEV map
is 128 x 200 x 176 — a dense 2-D problem MIOpen handles; nothing on it is a hot kernel of the path.

  spatial_features (B, C, H, W) -> NUM_CONTEXT_CONV x context block -> 1x1 conv + ReLU -> 1x1 conv -> logits
  context block (CONTEXT_CONV): 'separable' (default) = depthwise 3x3 + BN + ReLU, pointwise 1x1 + BN + ReLU — the
  dilated map is wide (128 channels x 35200 cells x batch) and mostly empty, a full 3x3 convolution over it would cost
  more FLOPs than the whole point backbone; 'full' = 3x3 conv + BN + ReLU as CenterPoint's shared_conv.
  batch_dict['bev_heatmap'] = sigmoid(logits)  (B, num_class, H, W)
"""
import os

import numpy as np
import torch
import torch.nn as nn

from ..utils import centernet_utils, loss_utils
from .point_head_template import _get
from ..fused_bn import TrainSequential


def fused_bn_enabled():
    from .. import fused_bn
    return fused_bn.ENABLED


class _Depthwise3x3CL(torch.autograd.Function):
    """Depthwise 3x3 convolution (padding 1, no bias) on a channels-last fp32 map, forward and backward on the HIP
    kernels of csrc/bev_head.hip.  MIOpen's depthwise path for this shape (128 channels x 200 x 176 x 32) costs ~170 ms
    per training step in its weight-gradient kernel alone; these are three memory-bound passes over the map.
    out_bf16 (training under bf16 autocast): fp32 arithmetic on the fp32 map, the OUTPUT rounded once to bf16 — what autocast
    gives a convolution's output, without its rounding of the inputs — and the gradient taken in bf16: the BatchNorm behind it
    and the two gradient kernels then move half the bytes (576 -> 288 MB per pass at bs = 32)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, out_bf16=False):
        from .. import _native
        rows = x.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            rows = rows.contiguous()
        B, H, W, C = rows.shape
        w9 = weight.reshape(C, 9).t().contiguous()                 # (9, C) tap-major
        out = torch.empty_like(rows, dtype=torch.bfloat16) if out_bf16 else torch.empty_like(rows)
        zero = torch.zeros(C, dtype=torch.float32, device=x.device)
        _native.call("pdm_bev_depthwise3x3_t", torch.cuda.current_stream(x.device).cuda_stream, B, H, W, C, rows.data_ptr(), 0,
                     w9.data_ptr(), zero.data_ptr(), out.data_ptr(), 1 if out_bf16 else 0, 0)
        ctx.save_for_backward(rows, w9)
        ctx.out_bf16 = bool(out_bf16)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        from .. import _native
        rows, w9 = ctx.saved_tensors
        B, H, W, C = rows.shape
        gb = 1 if ctx.out_bf16 else 0
        gr = (g.to(torch.bfloat16) if gb else g.float()).permute(0, 2, 3, 1)
        if not gr.is_contiguous():
            gr = gr.contiguous()
        stream = torch.cuda.current_stream(rows.device).cuda_stream
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx_rows = torch.empty_like(rows)
            zero = torch.zeros(C, dtype=torch.float32, device=rows.device)
            _native.call("pdm_bev_depthwise3x3_t", stream, B, H, W, C, gr.data_ptr(), gb, w9.flip(0).contiguous().data_ptr(),
                         zero.data_ptr(), gx_rows.data_ptr(), 0, 0)
            gx = gx_rows.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw9 = torch.zeros((9, C), dtype=torch.float32, device=rows.device)
            _native.call("pdm_bev_depthwise3x3_wgrad_t", stream, B, H, W, C, rows.data_ptr(), gr.data_ptr(), gb, gw9.data_ptr())
            gw = gw9.t().reshape(C, 1, 3, 3)
        return gx, gw, None


class _DepthwiseConv3x3(nn.Conv2d):
    """nn.Conv2d(c, c, 3, padding=1, groups=c, bias=False) — same parameters and state_dict — whose CUDA path is the
    HIP pair above."""

    def forward(self, x):
        if x.is_cuda and x.shape[1] % 4 == 0:
            from .. import fused_bn
            bf = self.training and fused_bn.ENABLED and fused_bn._bf16_autocast() and x.shape[1] % 8 == 0 and DEPTHWISE_BF16_OUT
            return _Depthwise3x3CL.apply(x, self.weight, bool(bf))
        return super().forward(x)


# 1 (default): under bf16 autocast in training the depthwise convolution hands on a bf16 map (see _Depthwise3x3CL)
DEPTHWISE_BF16_OUT = os.environ.get("PDM_DEPTHWISE_BF16_OUT", "1") == "1"


class PDMHeatmapHead(nn.Module):
    def __init__(self, model_cfg, input_channels, num_class, class_names=None, grid_size=None, point_cloud_range=None,
                 voxel_size=None, predict_boxes_when_training=False, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.class_names = class_names
        self.point_cloud_range = point_cloud_range
        self.voxel_size = voxel_size
        self.feature_map_stride = _get(_get(model_cfg, 'TARGET_ASSIGNER_CONFIG', {}), 'FEATURE_MAP_STRIDE', 8)
        width = _get(model_cfg, 'SHARED_CONV_CHANNEL', 64)
        layers, c = [], input_channels
        kind = _get(model_cfg, 'CONTEXT_CONV', 'separable')
        for _ in range(_get(model_cfg, 'NUM_CONTEXT_CONV', 1)):   # "context learning" over the dilated, mostly empty map
            if kind == 'separable':
                layers += [_DepthwiseConv3x3(c, c, 3, padding=1, groups=c, bias=False), nn.BatchNorm2d(c), nn.ReLU(),
                           nn.Conv2d(c, width, 1, bias=False), nn.BatchNorm2d(width), nn.ReLU()]
            else:
                layers += [nn.Conv2d(c, width, 3, padding=1, bias=False), nn.BatchNorm2d(width), nn.ReLU()]
            c = width
        self.shared_conv = TrainSequential(*layers)
        # (a TrainSequential: an nn.Sequential — same state_dict keys — whose 1x1 convolutions run on the bf16 MFMA row
        # kernels in training, fused_bn.py)
        self.hm = TrainSequential(nn.Conv2d(c, width, 1, bias=True), nn.ReLU(), nn.Conv2d(width, num_class, 1, bias=True))
        self.hm[-1].bias.data.fill_(-2.19)   # CenterPoint's prior: sigmoid(-2.19) = 0.1 (center_head.py:47)
        self.add_module('hm_loss_func', loss_utils.FocalLossCenterNet())
        self.forward_ret_dict = {}

    @staticmethod
    def sigmoid(x):
        return torch.clamp(x.sigmoid(), min=1e-4, max=1 - 1e-4)   # center_head.py:232

    def assign_targets(self, gt_boxes, feature_map_size):
        """gt_boxes (B, M, 8) [box7, class in 1..num_class; zero rows = padding], feature_map_size (H, W) ->
        heatmap (B, num_class, H, W): per box a gaussian of radius max(gaussian_radius(dx, dy cells), MIN_RADIUS)
        around the cell of its centre, max-merged (center_head.py:106-160), built on the device in one pass."""
        cfg = _get(self.model_cfg, 'TARGET_ASSIGNER_CONFIG', {})
        H, W = feature_map_size
        B, M, _ = gt_boxes.shape
        vx, vy = float(self.voxel_size[0]), float(self.voxel_size[1])
        s = self.feature_map_stride
        if gt_boxes.is_cuda and gt_boxes.dtype == torch.float32 and getattr(self, 'use_fused_loss', True) and _get(cfg, 'MAX_RADIUS', 8) <= 64:
            # one launch (pdm_heatmap_targets): the same arithmetic in the same order, gaussians max-merged by atomics
            from .. import heatmap_loss
            return heatmap_loss.heatmap_targets(gt_boxes, self.num_class, H, W, float(self.point_cloud_range[0]), float(self.point_cloud_range[1]),
                                                vx, vy, s, _get(cfg, 'GAUSSIAN_OVERLAP', 0.1), _get(cfg, 'MIN_RADIUS', 2), _get(cfg, 'MAX_RADIUS', 8))
        cx = torch.clamp((gt_boxes[..., 0] - self.point_cloud_range[0]) / vx / s, min=0, max=W - 0.5)
        cy = torch.clamp((gt_boxes[..., 1] - self.point_cloud_range[1]) / vy / s, min=0, max=H - 0.5)
        centers_int = torch.stack((cx, cy), dim=-1).int().long()
        dx, dy = gt_boxes[..., 3] / vx / s, gt_boxes[..., 4] / vy / s
        valid = (dx > 0) & (dy > 0) & (gt_boxes[..., 7] >= 1)
        safe = torch.where(valid, dx, torch.ones_like(dx)), torch.where(valid, dy, torch.ones_like(dy))
        radius = centernet_utils.gaussian_radius(safe[0], safe[1], min_overlap=_get(cfg, 'GAUSSIAN_OVERLAP', 0.1))
        radius = torch.clamp_min(radius.int(), min=_get(cfg, 'MIN_RADIUS', 2)).long()
        cls_idx = torch.stack((torch.arange(B, device=gt_boxes.device)[:, None].expand(B, M),
                               (gt_boxes[..., 7].long() - 1).clamp(min=0)), dim=-1)
        heatmap = gt_boxes.new_zeros((B, self.num_class, H, W))
        return centernet_utils.draw_gaussians(heatmap, cls_idx, centers_int, radius, valid,
                                              max_radius=_get(cfg, 'MAX_RADIUS', 8))

    def get_loss(self, tb_dict=None):
        tb_dict = {} if tb_dict is None else tb_dict
        logits, target = self.forward_ret_dict['hm_logits'], self.forward_ret_dict['heatmap']
        w = _get(_get(_get(self.model_cfg, 'LOSS_CONFIG'), 'LOSS_WEIGHTS'), 'cls_weight', 1.0)
        if (logits.is_cuda and logits.dtype in (torch.float32, torch.bfloat16) and target.dtype == torch.float32 and target.is_contiguous()
                and getattr(self, 'use_fused_loss', True) and type(self.hm_loss_func) is loss_utils.FocalLossCenterNet):
            # clamped sigmoid + penalty-reduced focal loss + its gradient: pdm_heatmap_focal_loss (two launches, bit-reproducible)
            from .. import heatmap_loss
            hm_loss = heatmap_loss.heatmap_focal_loss(logits, target, w)
        else:
            pred = self.sigmoid(logits.float())
            hm_loss = self.hm_loss_func(pred, target) * w
        tb_dict['hm_loss'] = hm_loss.detach()
        return hm_loss, tb_dict

    def _fused_logits(self, x):
        """Inference on the neck's channels-last grid: depthwise 3x3 + BN + ReLU as one HIP kernel
        (pdm_bev_depthwise3x3), then pointwise 1x1 + BN + ReLU -> 1x1 + ReLU -> 1x1 as ONE per-cell MLP on the MFMA row
        kernels (pdm_rows_mlp_fused, BatchNorm folded) — the map is read twice and written once instead of ten passes
        of MIOpen kernels.  None when the module's shape or the tensor's layout does not fit."""
        from .. import _native, fused
        mods = list(self.shared_conv)
        if (self.training or torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32 or len(mods) != 6
                or not isinstance(mods[0], nn.Conv2d) or mods[0].groups != mods[0].in_channels
                or not getattr(self, 'use_fused', True) or x.shape[1] % 16 != 0):
            return None
        rows = x.permute(0, 2, 3, 1)                       # (B, H, W, C): the storage itself for the neck's grid
        if not rows.is_contiguous():
            rows = rows.contiguous()
        B, H, W, C = rows.shape
        cache = self.__dict__.setdefault('_pdm_fused_cache', {})
        key = (fused._state_key(self.shared_conv), str(x.device))
        if cache.get('dw', (None,))[0] != key:
            dw, bn = mods[0], mods[1]
            s = (bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps))
            w = (dw.weight.detach().double().reshape(C, 9) * s[:, None]).t().contiguous().float()      # (9, C) tap-major
            shift = (bn.bias.detach().double() - bn.running_mean.detach().double() * s).float().contiguous()
            cache['dw'] = (key, w.to(x.device), shift.to(x.device))
        _, w, shift = cache['dw']
        pk = fused.cached_layers(self, 'pw', self, lambda: [(mods[3], mods[4]), (self.hm[0], None), (self.hm[2], None)], x.device)
        out = torch.empty((B, H, W, (self.num_class + 3) // 4 * 4), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if getattr(self, 'use_one_kernel', True) and pk.dims == [128, 64, 64, 16]:
            # the depthwise stage as the prologue of the per-cell MLP: the map is read once, nothing in between is written
            fused._count(f"pdm_bev_head_fused", B * H * W, pk)
            _native.call("pdm_bev_head_fused", stream, B, H, W, C, rows.data_ptr(), w.data_ptr(), shift.data_ptr(), pk.nlayers,
                         pk.dims_ptr, pk.wpack.data_ptr(), pk.bias.data_ptr(), 0, out.data_ptr(), out.shape[-1], self.num_class)
        else:
            mid = torch.empty_like(rows)
            _native.call("pdm_bev_depthwise3x3", stream, B, H, W, C, rows.data_ptr(), w.data_ptr(), shift.data_ptr(),
                         mid.data_ptr(), 1)
            fused.rows_forward(pk, mid, out, relu_last=False)
        return out[..., :self.num_class].permute(0, 3, 1, 2)

    def forward(self, data_dict):
        x = data_dict['spatial_features_2d'] if 'spatial_features_2d' in data_dict else data_dict['spatial_features']
        logits = self._fused_logits(x)
        if logits is not None:
            self.forward_ret_dict['hm_logits'] = logits
            data_dict['bev_heatmap'] = self.sigmoid(logits)
            return data_dict
        if x.dim() == 4 and not x.is_contiguous() and x.permute(0, 2, 3, 1).is_contiguous():
            x = x.contiguous(memory_format=torch.channels_last)   # the neck's grid IS channels-last storage: no copy
        if self.training and x.is_cuda and fused_bn_enabled():
            # the two stacks as ONE (hm(shared_conv(x)) is a plain chain): the BatchNorm + ReLU that ends shared_conv then rides in
            # hm's first contraction like an inner layer's, instead of an apply pass over the 72 M-element map and its two backward passes
            logits = TrainSequential._run(x, list(self.shared_conv) + list(self.hm))
        else:
            logits = self.hm(self.shared_conv(x))
        self.forward_ret_dict['hm_logits'] = logits
        if self.training:
            self.forward_ret_dict['heatmap'] = self.assign_targets(data_dict['gt_boxes'], logits.shape[2:])
        data_dict['bev_heatmap'] = self.sigmoid(logits)
        return data_dict
