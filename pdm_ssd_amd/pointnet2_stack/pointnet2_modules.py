"""Module API of the PointNet++ STACK layers (ragged batches) on MI355X: `StackSAModuleMSG`, `StackPointnetFPModule`,
`build_local_aggregation_module` with the constructor signatures, state_dict keys and forward contracts of
/root/reference/pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:10-160, and the vector-pool modules of the same
file (:160-470): `VectorPoolLocalInterpolateModule`, `VectorPoolAggregationModule`, `VectorPoolAggregationModuleMSG`.

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
from ..fused_bn import TrainSequential


def _cfg(config, key, default=None):
    if isinstance(config, dict):
        return config.get(key, default)
    return getattr(config, key, default)


def build_local_aggregation_module(input_channels, config):
    """ref :10-27 (StackSAModuleMSG branch; like the reference it prepends input_channels to config.MLPS in place)."""
    name = _cfg(config, 'NAME', 'StackSAModuleMSG')
    if name == 'VectorPoolAggregationModuleMSG':
        return VectorPoolAggregationModuleMSG(input_channels=input_channels, config=config), _cfg(config, 'MSG_POST_MLPS')[-1]
    if name != 'StackSAModuleMSG':
        raise NotImplementedError(name)
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
    return TrainSequential(*layers)


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


# ---- vector-pool aggregation (PV-RCNN++), ref :160-470 ------------------------------------------------------------

def _conv1d_bn_relu(widths, groups=1):
    layers = []
    for cin, cout in zip(widths[:-1], widths[1:]):
        layers += [nn.Conv1d(cin, cout, kernel_size=1, groups=groups, bias=False), nn.BatchNorm1d(cout), nn.ReLU()]
    return TrainSequential(*layers)


class VectorPoolLocalInterpolateModule(nn.Module):
    """ref :160-245 — features at the centres of every key point's local lattice cells, by inverse-distance
    interpolation over the three nearest of the key point's neighbours (three_nn_for_vector_pool_by_two_step), with
    the three offsets (9 values) appended when use_xyz.  state_dict: mlp.{3k}.weight / mlp.{3k+1}.* when mlp is given."""

    def __init__(self, mlp, num_voxels, max_neighbour_distance, nsample, neighbor_type, use_xyz=True,
                 neighbour_distance_multiplier=1.0, xyz_encoding_type='concat'):
        super().__init__()
        self.num_voxels = num_voxels
        self.num_total_grids = num_voxels[0] * num_voxels[1] * num_voxels[2]
        self.max_neighbour_distance = max_neighbour_distance
        self.neighbor_distance_multiplier = neighbour_distance_multiplier
        self.nsample = nsample
        self.neighbor_type = neighbor_type          # 1: ball, anything else: cube
        self.use_xyz = use_xyz
        self.xyz_encoding_type = xyz_encoding_type
        if mlp is not None:
            widths = list(mlp)                      # (upstream widens mlp[0] in the caller's list; a copy is widened here)
            if use_xyz and xyz_encoding_type == 'concat':
                widths[0] += 9
            self.mlp = _conv_bn_relu(widths)
        else:
            self.mlp = None
        self.num_avg_length_of_neighbor_idxs = 1000

    def forward(self, support_xyz, support_features, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt):
        """support_xyz (N,3), support_features (N,C), new_xyz (M,3), new_xyz_grid_centers (M,G,3)
        -> (M*G, C [+9]) or (M*G, mlp[-1])."""
        with torch.no_grad():
            dist, idx, avg_len = pointnet2_utils.three_nn_for_vector_pool_by_two_step(
                support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt, self.max_neighbour_distance,
                self.nsample, self.neighbor_type, self.num_avg_length_of_neighbor_idxs, self.num_total_grids,
                self.neighbor_distance_multiplier)
        self.num_avg_length_of_neighbor_idxs = max(self.num_avg_length_of_neighbor_idxs, avg_len.item())

        G = idx.shape[1]
        idx = idx.view(-1, 3)
        recip = 1.0 / (dist.view(-1, 3) + 1e-8)
        weight = recip / torch.clamp_min(recip.sum(dim=-1, keepdim=True), min=1e-8)
        empty = idx[:, 0] == -1                     # a cell whose key point had no neighbour at all
        idx = idx.masked_fill(empty[:, None], 0)

        feats = pointnet2_utils.three_interpolate(support_features, idx, weight)                     # (M*G, C)
        if self.use_xyz:
            if self.xyz_encoding_type != 'concat':
                raise NotImplementedError(self.xyz_encoding_type)
            offsets = new_xyz_grid_centers.reshape(-1, 1, 3) - support_xyz[idx.long()]                   # (M*G, 3, 3)
            feats = torch.cat((feats, offsets.reshape(-1, 9)), dim=-1)
        feats = feats.masked_fill(empty[:, None], 0)
        if self.mlp is not None:
            feats = self.mlp(feats.t()[None, :, :, None]).squeeze(0).squeeze(-1).t()
        assert feats.shape[0] == new_xyz.shape[0] * G
        return feats


class VectorPoolAggregationModule(nn.Module):
    """ref :247-420 — VectorPool aggregation of one scale: the support features (channel groups summed down to
    num_reduced_channels) are gathered per local lattice cell (interpolated at the cell centres, averaged over the
    cell, or taken from its first point), each cell gets its own linear map (a grouped 1x1 convolution), and shared
    MLPs follow.  state_dict: separate_local_aggregation_layer.{0,1}.*, post_mlps.{3k,3k+1}.*."""

    def __init__(self, input_channels, num_local_voxel=(3, 3, 3), local_aggregation_type='local_interpolation',
                 num_reduced_channels=30, num_channels_of_local_aggregation=32, post_mlps=(128,),
                 max_neighbor_distance=None, neighbor_nsample=-1, neighbor_type=0, neighbor_distance_multiplier=2.0):
        super().__init__()
        assert local_aggregation_type in ['local_interpolation', 'voxel_avg_pool', 'voxel_random_choice']
        self.num_local_voxel = num_local_voxel
        self.total_voxels = num_local_voxel[0] * num_local_voxel[1] * num_local_voxel[2]
        self.local_aggregation_type = local_aggregation_type
        self.input_channels = input_channels
        self.num_reduced_channels = input_channels if num_reduced_channels is None else num_reduced_channels
        self.num_channels_of_local_aggregation = num_channels_of_local_aggregation
        self.max_neighbour_distance = max_neighbor_distance
        self.neighbor_nsample = neighbor_nsample
        self.neighbor_type = neighbor_type

        if local_aggregation_type == 'local_interpolation':
            self.local_interpolate_module = VectorPoolLocalInterpolateModule(
                mlp=None, num_voxels=num_local_voxel, max_neighbour_distance=max_neighbor_distance, nsample=neighbor_nsample,
                neighbor_type=neighbor_type, neighbour_distance_multiplier=neighbor_distance_multiplier)
            per_cell = self.num_reduced_channels + 9
        else:
            self.local_interpolate_module = None
            per_cell = self.num_reduced_channels + 3
        num_c_out = self.total_voxels * num_channels_of_local_aggregation
        self.separate_local_aggregation_layer = _conv1d_bn_relu([per_cell * self.total_voxels, num_c_out], groups=self.total_voxels)
        self.post_mlps = _conv1d_bn_relu([num_c_out] + list(post_mlps))
        self.num_mean_points_per_grid = 20
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def extra_repr(self) -> str:
        return (f'radius={self.max_neighbour_distance}, local_voxels=({self.num_local_voxel}, '
                f'local_aggregation_type={self.local_aggregation_type}, '
                f'num_c_reduction={self.input_channels}->{self.num_reduced_channels}, '
                f'num_c_local_aggregation={self.num_channels_of_local_aggregation}')

    def vector_pool_with_voxel_query(self, xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt):
        """-> ((M, G * (3 + C)) per cell [mean offset, pooled features], point_cnt_of_grid (M, G))."""
        pooling_type = 0 if self.local_aggregation_type == 'voxel_avg_pool' else 1
        pooled, local_xyz, mean_pts, point_cnt_of_grid = pointnet2_utils.vector_pool_with_voxel_query_op(
            xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt, self.num_local_voxel[0], self.num_local_voxel[1],
            self.num_local_voxel[2], self.max_neighbour_distance, self.num_reduced_channels, 1, self.num_mean_points_per_grid,
            self.neighbor_nsample, self.neighbor_type, pooling_type)
        self.num_mean_points_per_grid = max(self.num_mean_points_per_grid, mean_pts.item())
        M = pooled.shape[0]
        per_cell = torch.cat((local_xyz.view(M, -1, 3), pooled.view(M, -1, self.num_reduced_channels)), dim=-1)
        return per_cell.view(M, -1), point_cnt_of_grid

    @staticmethod
    def get_dense_voxels_by_center(point_centers, max_neighbour_distance, num_voxels):
        """(N,3) -> (N, total_voxels, 3): the centres of the num_voxels lattice over the cube of half-width
        max_neighbour_distance around every point, x slowest / z fastest (the cell order of the pooling kernels)."""
        R = max_neighbour_distance
        axes = [torch.arange(n, device=point_centers.device, dtype=torch.float32) * (2 * R / n) + (R / n - R) for n in num_voxels]
        offsets = torch.stack(torch.meshgrid(*axes, indexing='ij'), dim=-1).view(-1, 3)
        return point_centers[:, None, :] + offsets[None, :, :]

    def vector_pool_with_local_interpolate(self, xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt):
        """-> (M, total_voxels * (C + 9))."""
        centres = self.get_dense_voxels_by_center(new_xyz, self.max_neighbour_distance, self.num_local_voxel)
        cell_feats = self.local_interpolate_module(
            support_xyz=xyz, support_features=features, xyz_batch_cnt=xyz_batch_cnt, new_xyz=new_xyz,
            new_xyz_grid_centers=centres, new_xyz_batch_cnt=new_xyz_batch_cnt)
        return cell_feats.contiguous().view(-1, self.total_voxels * cell_feats.shape[-1])

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, **kwargs):
        """xyz (N,3), features (N,C), new_xyz (M,3) -> (new_xyz, new_features (M, post_mlps[-1]))."""
        N, C = features.shape
        assert C % self.num_reduced_channels == 0, \
            f'the input channels ({C}) should be an integral multiple of num_reduced_channels({self.num_reduced_channels})'
        features = features.view(N, -1, self.num_reduced_channels).sum(dim=1)
        if self.local_aggregation_type == 'local_interpolation':
            vec = self.vector_pool_with_local_interpolate(xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt)
        else:
            vec, _ = self.vector_pool_with_voxel_query(xyz, xyz_batch_cnt, features.contiguous(), new_xyz, new_xyz_batch_cnt)
        out = self.post_mlps(self.separate_local_aggregation_layer(vec.t()[None, :, :]))
        return new_xyz, out.squeeze(0).t()


class VectorPoolAggregationModuleMSG(nn.Module):
    """ref :423-470 — several VectorPoolAggregationModule scales (config.GROUP_CFG_k), concatenated behind the key
    points' xyz and mixed by shared MLPs.  state_dict: layer_{k}.*, msg_post_mlps.{3k,3k+1}.*."""

    def __init__(self, input_channels, config):
        super().__init__()
        self.model_cfg = config
        self.num_groups = _cfg(config, 'NUM_GROUPS')
        c_in = 3                                                                    # the key points' xyz
        for k in range(self.num_groups):
            group = _cfg(config, f'GROUP_CFG_{k}')
            setattr(self, f'layer_{k}', VectorPoolAggregationModule(
                input_channels=input_channels, num_local_voxel=_cfg(group, 'NUM_LOCAL_VOXEL'), post_mlps=_cfg(group, 'POST_MLPS'),
                max_neighbor_distance=_cfg(group, 'MAX_NEIGHBOR_DISTANCE'), neighbor_nsample=_cfg(group, 'NEIGHBOR_NSAMPLE'),
                local_aggregation_type=_cfg(config, 'LOCAL_AGGREGATION_TYPE'),
                num_reduced_channels=_cfg(config, 'NUM_REDUCED_CHANNELS', None),
                num_channels_of_local_aggregation=_cfg(config, 'NUM_CHANNELS_OF_LOCAL_AGGREGATION'),
                neighbor_distance_multiplier=2.0))
            c_in += _cfg(group, 'POST_MLPS')[-1]
        self.msg_post_mlps = _conv1d_bn_relu([c_in] + list(_cfg(config, 'MSG_POST_MLPS')))

    def forward(self, **kwargs):
        outs = [getattr(self, f'layer_{k}')(**kwargs) for k in range(self.num_groups)]
        key_xyz = outs[-1][0]
        features = torch.cat([key_xyz] + [f for _, f in outs], dim=-1)
        return key_xyz, self.msg_post_mlps(features.t()[None, :, :]).squeeze(0).t()
