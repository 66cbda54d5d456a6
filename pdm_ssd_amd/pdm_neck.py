"""PDM neck — Point Dilation Mechanism in the MAP_TO_BEV slot of an OpenPCDet detector.

The reference snapshot ships no source for it (README.md:12 describes it in one
sentence), so this module implements the repository's own spec (DESIGN.md "PDM spec"):
  point dilation -> SH x Gaussian feature filling -> multi-centre scatter-add -> height compression.
Module contract = that of the reference's map_to_bev modules
(/root/reference/pcdet/models/backbones_2d/map_to_bev/height_compression.py:5-26,
 detector3d_template.py:85-95): cls(model_cfg=, grid_size=, ...), `.num_bev_features`,
forward(batch_dict) writes 'spatial_features' (B, C*D, H, W) and 'spatial_features_stride'.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused, pdm_ops
from .fused_bn import TrainSequential


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class PDMNeck(nn.Module):
    def __init__(self, model_cfg, grid_size=None, voxel_size=None, point_cloud_range=None,
                 input_channels=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.source_layer = _get(model_cfg, 'SOURCE_LAYER', 2)       # index into batch_dict['sa_xyz']
        self.feature_dim = _get(model_cfg, 'FEATURE_DIM', 128)        # C
        self.dilation = tuple(_get(model_cfg, 'DILATION', [7, 7, 1]))  # Kx, Ky, Kz
        self.degree = _get(model_cfg, 'SH_DEGREE', 2)
        self.stride = _get(model_cfg, 'BEV_STRIDE', 8)
        self.height_bins = _get(model_cfg, 'HEIGHT_BINS', 1)          # D
        self.sigma_min = _get(model_cfg, 'SIGMA_MIN', 0.2)
        self.normalize = _get(model_cfg, 'NORMALIZE', True)
        self.layout = 1 if _get(model_cfg, 'CHANNELS_LAST', True) else 0
        in_ch = input_channels if input_channels is not None else _get(model_cfg, 'INPUT_CHANNELS')
        assert in_ch is not None, 'PDMNeck needs INPUT_CHANNELS (channels of the source SA layer)'
        pcr = point_cloud_range
        cell = [voxel_size[0] * self.stride, voxel_size[1] * self.stride,
                (pcr[5] - pcr[2]) / self.height_bins]
        self.grid = pdm_ops.BevGrid(pcr, cell)
        assert self.grid.D == self.height_bins
        self.nsh = (self.degree + 1) ** 2
        self.proj = TrainSequential(nn.Conv1d(in_ch, self.feature_dim, 1, bias=False),
                                  nn.BatchNorm1d(self.feature_dim), nn.ReLU())
        self.coef = nn.Conv1d(in_ch, self.nsh + 1, 1)   # SH coefficients + raw scale
        self.num_bev_features = self.feature_dim * self.height_bins
        nn.init.zeros_(self.coef.weight)
        with torch.no_grad():
            self.coef.bias.zero_()
            self.coef.bias[0] = 1.0 / 0.28209479177387814  # start as a pure Gaussian (S == 1)

    def forward(self, batch_dict):
        xyz = batch_dict['sa_xyz'][self.source_layer]            # (B, P, 3)
        src = batch_dict['sa_features'][self.source_layer]       # (B, Cin, P)
        infer = not self.training and not torch.is_grad_enabled() and src.is_cuda and src.dtype == torch.float32
        if infer and self.feature_dim % 4 == 0 and getattr(self, 'use_fused', True):
            # the two 1x1 convolutions as per-row MLPs on the point-major rows (BN folded; no transposes)
            src_pm = src.transpose(1, 2).contiguous()            # free when src is a view of (B, P, Cin) storage
            pk = fused.cached_pack(self, 'proj', self.proj, src.device)
            pc = fused.cached_layers(self, 'coef', self.coef, lambda: [(self.coef, None)], src.device)
            feat = torch.empty((src_pm.shape[0], src_pm.shape[1], self.feature_dim), dtype=torch.float32, device=src.device)
            fused.rows_forward(pk, src_pm, feat, relu_last=True)
            co = torch.empty((src_pm.shape[0], src_pm.shape[1], pc.dims[-1]), dtype=torch.float32, device=src.device)
            fused.rows_forward(pc, src_pm, co, relu_last=False)
        else:
            feat = self.proj(src).transpose(1, 2).contiguous()   # (B, P, C)
            co = self.coef(src).transpose(1, 2)                  # (B, P, nsh+1)
        sh = co[..., :self.nsh].contiguous()
        sigma = F.softplus(co[..., self.nsh]) + self.sigma_min
        inv2s2 = (0.5 / (sigma * sigma)).contiguous()
        if not torch.is_grad_enabled() and self.layout == 1 and pdm_ops.gather_supported(self.feature_dim, self.grid.D) \
                and getattr(self, 'use_gather', True):
            # inference: deterministic gather form, normalisation fused, the grid is written exactly once
            grid, wsum = pdm_ops.pdm_gather(xyz, feat, sh, inv2s2, self.grid, self.dilation, self.degree,
                                            normalize=self.normalize)
            batch_dict['spatial_features'] = grid.permute(0, 3, 1, 2)
            batch_dict['spatial_features_stride'] = self.stride
            batch_dict['pdm_weight_sum'] = wsum
            return batch_dict
        if (self.normalize and self.layout == 1 and feat.is_cuda and pdm_ops.gather_supported(self.feature_dim, self.grid.D)
                and getattr(self, 'use_gather', True) and (feat.requires_grad or sh.requires_grad)):
            # training: the gather kernel forward (grid written once, normalised), the usual two kernels backward
            grid, wsum = pdm_ops.pdm_gather_normalized(xyz.contiguous(), feat, sh, inv2s2, self.grid, self.dilation, self.degree)
            batch_dict['spatial_features'] = grid.permute(0, 3, 1, 2)
            batch_dict['spatial_features_stride'] = self.stride
            batch_dict['pdm_weight_sum'] = wsum
            return batch_dict
        grid, wsum = pdm_ops.pdm_scatter(xyz.contiguous(), feat, sh, inv2s2, self.grid, self.dilation,
                                         self.degree, self.layout)
        if self.normalize:
            if torch.is_grad_enabled() and (feat.requires_grad or sh.requires_grad):
                B = grid.shape[0]
                if self.layout == 1:
                    grid = pdm_ops.bev_normalize(grid, wsum, self.feature_dim, self.grid)   # one kernel each way
                else:
                    w = wsum.permute(0, 3, 1, 2).unsqueeze(1)                 # (B,1,D,H,W)
                    g5 = grid.view(B, self.feature_dim, self.grid.D, self.grid.H, self.grid.W)
                    g5 = torch.where(w.abs() > 1e-6, g5 / torch.where(w.abs() > 1e-6, w, torch.ones_like(w)), g5)
                    grid = g5.view_as(grid)
            else:
                pdm_ops.bev_normalize_(grid, wsum, self.feature_dim, self.grid, self.layout)
        # height compression (height_compression.py:21-23): (B, C, D, H, W) viewed as (B, C*D, H, W)
        batch_dict['spatial_features'] = grid.permute(0, 3, 1, 2) if self.layout == 1 else grid
        batch_dict['spatial_features_stride'] = self.stride
        batch_dict['pdm_weight_sum'] = wsum
        return batch_dict
