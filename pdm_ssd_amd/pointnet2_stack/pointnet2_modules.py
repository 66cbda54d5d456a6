"""Module API of the PointNet++ STACK layers (ragged batches) on MI355X: `StackSAModuleMSG`, `StackPointnetFPModule`,
`build_local_aggregation_module` with the constructor signatures, state_dict keys and forward contracts of
/root/reference/pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:10-160.  The vector-pool modules of that file
are outside this build.

In eval mode without autograd both modules run through the fused fp32-MFMA kernels of the batch path
(pdm_ssd_amd/fused.py): a stacked batch is handed over as ONE sample whose neighbour indices were made global.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from .. import fused
from ..pointnet2_batch.pointnet2_modules import PRE_MIN_CIN


def _cfg(config, key, default=None):
    if isinstance(config, dict):
        return config.get(key, default)
    return getattr(config, key, default)


def build_local_aggregation_module(input_channels, config):
    """ref :10-27 (StackSAModuleMSG branch; like the reference it prepends input_channels to config.MLPS in place)."""
    name = _cfg(config, 'NAME', 'StackSAModuleMSG')
    if name != 'StackSAModuleMSG':
        raise NotImplementedError(f'{name}: only StackSAModuleMSG is built (vector pooling is out of scope)')
    mlps = _cfg(config, 'MLPS')
    for k in range(len(mlps)):
        mlps[k] = [input_channels] + mlps[k]
    layer = StackSAModuleMSG(radii=_cfg(config, 'POOL_RADIUS'), nsamples=_cfg(config, 'NSAMPLE'), mlps=mlps,
                             use_xyz=True, pool_method='max_pool')
    return layer, sum(x[-1] for x in mlps)


def _conv_bn_relu(spec):
    layers = []
    for cin, cout in zip(spec[:-1], spec[1:]):
        layers += [nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU()]
    return nn.Sequential(*layers)


def _row_starts(cnt_src, cnt_rows):
    """start offset (in the stacked source) of the sample each stacked row belongs to, int32 (rows,)."""
    starts = torch.cumsum(cnt_src.long(), 0) - cnt_src.long()
    return torch.repeat_interleave(starts, cnt_rows.long()).int()


class StackSAModuleMSG(nn.Module):
    """ref :30-115.  Like the reference the constructor adds 3 to mlps[i][0] IN PLACE when use_xyz and applies
    kaiming-normal / unit-BN initialisation (:68-76)."""

    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(_conv_bn_relu(spec))
        self.pool_method = pool_method
        self.use_xyz = use_xyz
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def _fused_ok(self, xyz, features):
        return (not self.training and not torch.is_grad_enabled() and getattr(self, 'use_fused', True)
                and self.pool_method == 'max_pool' and self.use_xyz and xyz.is_cuda and xyz.dtype == torch.float32
                and (features is None or features.dtype == torch.float32)
                and all(g.nsample % 16 == 0 for g in self.groupers))

    @torch.no_grad()
    def _forward_fused(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features):
        cin = 0 if features is None else features.shape[1]
        perm = list(range(3, 3 + cin)) + [0, 1, 2]      # reference order [xyz, features] -> kernel order [features, xyz]
        packs = [fused.cached_pack(self, i, mlp, xyz.device, perm) for i, mlp in enumerate(self.mlps)]
        if any(pk is None or pk.cin != cin + 3 or pk.cout % 4 for pk in packs):
            return None
        xyz1, new1 = xyz.contiguous()[None], new_xyz.contiguous()[None]
        feat1 = None if features is None else features.contiguous()[None]        # (1, N, C): already point-major
        M = new_xyz.shape[0]
        out = torch.empty((1, M, sum(pk.cout for pk in packs)), dtype=torch.float32, device=xyz.device)
        starts = _row_starts(xyz_batch_cnt, new_xyz_batch_cnt)
        pre = None
        if cin >= PRE_MIN_CIN and getattr(self, 'use_pre', True):
            pre = fused.cached_pre_packs(self, 'pre', list(self.mlps), xyz.device, range(3, 3 + cin), range(3))
        if pre is not None:
            prepack, packs = pre
            z = torch.empty((1, xyz.shape[0], prepack.width), dtype=torch.float32, device=xyz.device)
            fused.rows_forward(prepack, feat1, z, relu_last=False)
        coff = 0
        for i, (grouper, pk, mlp) in enumerate(zip(self.groupers, packs, self.mlps)):
            idx, empty = pointnet2_utils.ball_query(grouper.radius, grouper.nsample, xyz, xyz_batch_cnt, new_xyz,
                                                    new_xyz_batch_cnt)
            gidx = (idx + starts[:, None])[None].contiguous()                     # global neighbour indices
            if pre is not None:
                fused.sa_scale_forward_pre(pk, xyz1, new1, z, prepack.offsets[i], gidx, out, coff)
            else:
                fused.sa_scale_forward(pk, xyz1, new1, feat1, gidx, out, coff)
            # the reference zeroes the grouped inputs of empty balls (pointnet2_utils.py:147-151): their output is the
            # network's response to an all-zero group
            zero_in = torch.zeros((1, mlp[0].in_channels, 1, 1), dtype=torch.float32, device=xyz.device)
            out[0, :, coff:coff + pk.cout] = torch.where(empty[:, None], mlp(zero_in).view(1, -1),
                                                         out[0, :, coff:coff + pk.cout])
            coff += pk.cout
        return out[0]

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True):
        """xyz (N1+N2+...,3), new_xyz (M1+M2+...,3), features (N1+N2+...,C) -> (new_xyz, (M1+M2+..., sum_k mlps[k][-1]))."""
        if self._fused_ok(xyz, features):
            out = self._forward_fused(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)
            if out is not None:
                return new_xyz, out
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            grouped, _ = grouper(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)   # (M, C, nsample)
            x = mlp(grouped.permute(1, 0, 2).unsqueeze(0))                                    # (1, C', M, nsample)
            if self.pool_method == 'max_pool':
                x = F.max_pool2d(x, kernel_size=[1, x.size(3)])
            elif self.pool_method == 'avg_pool':
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)])
            else:
                raise NotImplementedError
            pooled.append(x.squeeze(-1).squeeze(0).permute(1, 0))                             # (M, C')
        return new_xyz, torch.cat(pooled, dim=1)


class StackPointnetFPModule(nn.Module):
    """ref :118-160."""

    def __init__(self, *, mlp: List[int]):
        super().__init__()
        self.mlp = _conv_bn_relu(mlp)

    def forward(self, unknown, unknown_batch_cnt, known, known_batch_cnt, unknown_feats=None, known_feats=None):
        """unknown (N,3), known (M,3), unknown_feats (N,C1)|None, known_feats (M,C2) -> (N, mlp[-1])."""
        dist, idx = pointnet2_utils.three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt)
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / torch.sum(dist_recip, dim=-1, keepdim=True)
        if not self.training and not torch.is_grad_enabled() and getattr(self, 'use_fused', True) \
                and known_feats.is_cuda and known_feats.dtype == torch.float32:
            pk = fused.cached_pack(self, 0, self.mlp, known_feats.device)
            cs = 0 if unknown_feats is None else unknown_feats.shape[1]
            if pk is not None and pk.cin == known_feats.shape[1] + cs and pk.cout % 4 == 0:
                out = torch.empty((1, unknown.shape[0], pk.cout), dtype=torch.float32, device=known_feats.device)
                fused.fp_forward(pk, known_feats.contiguous()[None], None if unknown_feats is None else
                                 unknown_feats.float().contiguous()[None], idx[None].contiguous(),
                                 weight[None].contiguous(), out)
                return out[0]
        interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        x = interpolated if unknown_feats is None else torch.cat([interpolated, unknown_feats], dim=1)
        return self.mlp(x.permute(1, 0)[None, :, :, None]).squeeze(0).squeeze(-1).permute(1, 0)
