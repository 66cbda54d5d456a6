"""Voxel-window neighbour query on MI355X: the operator API of
/root/reference/pcdet/ops/pointnet2/pointnet2_stack/voxel_query_utils.py (VoxelQuery, voxel_query,
VoxelQueryAndGrouping) over the HIP kernel behind voxel_query_wrapper.  No PyTorch or CPU fallback.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_stack_hip as pointnet2
from . import pointnet2_utils


class VoxelQuery(Function):
    """ref :10-44 — for every key point, the points recorded in the voxels of a (2*z_range+1, 2*y_range+1, 2*x_range+1)
    window around its voxel that lie within `radius` of it, in window order (z outer, x inner), the first hit
    filling the row.  -> (idx (M, nsample) int32 into the STACKED xyz, empty_ball_mask (M,) bool)."""

    @staticmethod
    def forward(ctx, max_range, radius: float, nsample: int, xyz: torch.Tensor,
                new_xyz: torch.Tensor, new_coords: torch.Tensor, point_indices: torch.Tensor):
        for name, t in (("new_xyz", new_xyz), ("xyz", xyz), ("new_coords", new_coords), ("point_indices", point_indices)):
            assert t.is_contiguous(), f"{name} must be contiguous"
        M = new_coords.shape[0]
        B, Z, Y, X = point_indices.shape
        idx = torch.zeros((M, nsample), dtype=torch.int32, device=xyz.device)   # ref :34
        z_range, y_range, x_range = max_range
        pointnet2.voxel_query_wrapper(M, Z, Y, X, nsample, radius, z_range, y_range, x_range, new_xyz.float(), xyz.float(),
                                      pointnet2_utils._i32(new_coords), pointnet2_utils._i32(point_indices), idx)
        empty_ball_mask = (idx[:, 0] == -1)   # ref :40-41
        idx[empty_ball_mask] = 0
        ctx.mark_non_differentiable(idx, empty_ball_mask)
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return (None,) * 7


voxel_query = VoxelQuery.apply


class VoxelQueryAndGrouping(nn.Module):
    """Voxel-window query followed by the stacked grouping operator (ref :50-101)."""

    def __init__(self, max_range, radius: float, nsample: int):
        """max_range: (z, y, x) half-widths of the voxel window; radius / nsample as in the ball query."""
        super().__init__()
        self.max_range, self.radius, self.nsample = max_range, radius, nsample

    def forward(self, new_coords: torch.Tensor, xyz: torch.Tensor, xyz_batch_cnt: torch.Tensor,
                new_xyz: torch.Tensor, new_xyz_batch_cnt: torch.Tensor,
                features: torch.Tensor, voxel2point_indices: torch.Tensor):
        """new_coords (M1+M2.., 4) [batch, z, y, x] voxel coordinates of the key points; xyz (N1+N2.., 3); features
        (N1+N2.., C); voxel2point_indices (B, Z, Y, X) index of the point recorded in each voxel or -1.
        -> (grouped_features (M, C, nsample), grouped_xyz (M, 3, nsample), empty_ball_mask (M,)).
        As upstream, every sample must hold the same number of key points (its (batch_size, -1, nsample) view, ref :85)."""
        n_pts, n_keys = int(xyz_batch_cnt.sum()), int(new_xyz_batch_cnt.sum())
        assert xyz.shape[0] == n_pts, f'xyz: {tuple(xyz.shape)}, xyz_batch_cnt sums to {n_pts}'
        assert new_coords.shape[0] == n_keys, f'new_coords: {tuple(new_coords.shape)}, new_xyz_batch_cnt sums to {n_keys}'
        B = xyz_batch_cnt.shape[0]
        assert n_keys % B == 0, 'the samples must hold equally many key points'

        found, nothing_found = voxel_query(self.max_range, self.radius, self.nsample, xyz, new_xyz, new_coords, voxel2point_indices)
        # the query answers with indices into the stacked xyz; the grouping operator wants them per sample (ref :85-91):
        # one broadcast subtraction of each sample's first index instead of a python loop over samples
        first_of_sample = (torch.cumsum(xyz_batch_cnt, 0) - xyz_batch_cnt).to(found.dtype)
        local = (found.view(B, n_keys // B, self.nsample) - first_of_sample.view(B, 1, 1)).view(n_keys, self.nsample)
        local[nothing_found] = 0

        grouped = [pointnet2_utils.grouping_operation(t, xyz_batch_cnt, local, new_xyz_batch_cnt) for t in (features, xyz)]
        return grouped[0], grouped[1], nothing_found
