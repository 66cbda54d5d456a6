"""Operator API of the PointNet++ STACK ops (ragged batches) on MI355X.

Mirrors the public names, argument order, shapes, dtypes and zero-fill behaviour of
/root/reference/pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py for the PointNet++ operators
(ball_query, grouping_operation, QueryAndGroup, farthest_point_sample, stack_farthest_point_sample, three_nn,
three_interpolate) and for the vector-pool operators (three_nn_for_vector_pool_by_two_step,
vector_pool_with_voxel_query_op); the voxel query lives in voxel_query_utils.py as upstream.  Every operator
dispatches to a hand-written HIP kernel in libpdmssd_hip.so; there is no PyTorch or CPU fallback.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_stack_hip as pointnet2


def _i32(t):
    return t if t.dtype == torch.int32 else t.int()


class BallQuery(Function):
    """ref :8-47 — (idx (M,nsample) int32 local to the sample, empty_ball_mask (M,) bool); no gradient."""

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, xyz_batch_cnt: torch.Tensor,
                new_xyz: torch.Tensor, new_xyz_batch_cnt):
        for name, t in (("new_xyz", new_xyz), ("new_xyz_batch_cnt", new_xyz_batch_cnt), ("xyz", xyz),
                        ("xyz_batch_cnt", xyz_batch_cnt)):
            assert t.is_contiguous(), f"{name} must be contiguous"
        B = xyz_batch_cnt.shape[0]
        M = new_xyz.shape[0]
        idx = torch.zeros((M, nsample), dtype=torch.int32, device=xyz.device)   # ref :33
        pointnet2.ball_query_wrapper(B, M, radius, nsample, new_xyz.float(), _i32(new_xyz_batch_cnt), xyz.float(),
                                     _i32(xyz_batch_cnt), idx)
        empty_ball_mask = (idx[:, 0] == -1)   # ref :36-37
        idx[empty_ball_mask] = 0
        ctx.mark_non_differentiable(idx)
        ctx.mark_non_differentiable(empty_ball_mask)
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None


ball_query = BallQuery.apply


class GroupingOperation(Function):
    """ref :52-106 — features (N,C), idx (M,nsample) -> (M,C,nsample); backward scatters with atomics."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, features_batch_cnt: torch.Tensor,
                idx: torch.Tensor, idx_batch_cnt: torch.Tensor):
        for name, t in (("features", features), ("features_batch_cnt", features_batch_cnt), ("idx", idx),
                        ("idx_batch_cnt", idx_batch_cnt)):
            assert t.is_contiguous(), f"{name} must be contiguous"
        assert features.shape[0] == int(features_batch_cnt.sum()), \
            f"{features.shape[0]} feature rows but features_batch_cnt sums to {int(features_batch_cnt.sum())}"
        assert idx.shape[0] == int(idx_batch_cnt.sum()), \
            f"{idx.shape[0]} index rows but idx_batch_cnt sums to {int(idx_batch_cnt.sum())}"
        M, nsample = idx.size()
        N, C = features.size()
        B = idx_batch_cnt.shape[0]
        output = torch.empty((M, C, nsample), dtype=torch.float32, device=features.device)
        pointnet2.group_points_wrapper(B, M, C, nsample, features.float(), _i32(features_batch_cnt), idx,
                                       _i32(idx_batch_cnt), output)
        ctx.for_backwards = (B, N, idx, features_batch_cnt, idx_batch_cnt)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        B, N, idx, features_batch_cnt, idx_batch_cnt = ctx.for_backwards
        M, C, nsample = grad_out.size()
        grad_features = torch.zeros((N, C), dtype=torch.float32, device=grad_out.device)   # ref :100
        pointnet2.group_points_grad_wrapper(B, M, C, N, nsample, grad_out.float().contiguous(), idx,
                                            _i32(idx_batch_cnt), _i32(features_batch_cnt), grad_features)
        return grad_features, None, None, None


grouping_operation = GroupingOperation.apply


class QueryAndGroup(nn.Module):
    """ref :112-159 — ball query, grouped xyz minus centre, grouped features, concat; empty balls zeroed."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, xyz_batch_cnt: torch.Tensor,
                new_xyz: torch.Tensor, new_xyz_batch_cnt: torch.Tensor,
                features: torch.Tensor = None):
        assert xyz.shape[0] == int(xyz_batch_cnt.sum()), f"{xyz.shape[0]} points, xyz_batch_cnt {xyz_batch_cnt.tolist()}"
        assert new_xyz.shape[0] == int(new_xyz_batch_cnt.sum()), \
            f"{new_xyz.shape[0]} centres, new_xyz_batch_cnt {new_xyz_batch_cnt.tolist()}"
        idx, empty_ball_mask = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)  # (M, 3, nsample)
        grouped_xyz = grouped_xyz - new_xyz.unsqueeze(-1)
        grouped_xyz[empty_ball_mask] = 0
        if features is not None:
            grouped_features = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)  # (M, C, nsample)
            grouped_features[empty_ball_mask] = 0
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "QueryAndGroup needs features or use_xyz=True"
            new_features = grouped_xyz
        return new_features, idx


class FarthestPointSampling(Function):
    """ref :162-188 — the batch kernel on (B,N,3): identical to pointnet2_batch's operator."""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int):
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.farthest_point_sampling_wrapper(B, N, npoint, xyz.float(), temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class StackFarthestPointSampling(Function):
    """ref :191-224 — xyz (N1+N2+...,3), per-sample counts, npoint (int | list | tensor) -> GLOBAL int32 indices
    packed per sample."""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, npoint):
        assert xyz.is_contiguous() and xyz.shape[1] == 3
        if not torch.is_tensor(npoint):   # an int applies to every sample (ref :205-208)
            per_sample = list(npoint) if isinstance(npoint, (list, tuple)) else [int(npoint)] * len(xyz_batch_cnt)
            npoint = torch.tensor(per_sample, dtype=torch.int32, device=xyz.device)
        N, _ = xyz.size()
        temp = torch.full((N,), 1e10, dtype=torch.float32, device=xyz.device)
        output = torch.empty((int(npoint.sum().item()),), dtype=torch.int32, device=xyz.device)
        pointnet2.stack_farthest_point_sampling_wrapper(xyz.float(), temp, _i32(xyz_batch_cnt).contiguous(), output,
                                                        _i32(npoint).contiguous())
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


stack_farthest_point_sample = StackFarthestPointSampling.apply


class ThreeNN(Function):
    """ref :228-259 — (dist (N,3) = sqrt of the squared distances, idx (N,3) int32 into the stacked known set)."""

    @staticmethod
    def forward(ctx, unknown, unknown_batch_cnt, known, known_batch_cnt):
        assert unknown.dim() == 2 and unknown.shape[1] == 3, "unknown must be (N1+N2+..., 3)"
        assert known.dim() == 2 and known.shape[1] == 3, "known must be (M1+M2+..., 3)"
        assert len(unknown_batch_cnt) == len(known_batch_cnt), "one count per sample on both sides"
        dist2 = unknown.new_zeros(unknown.shape, dtype=torch.float32)
        idx = torch.zeros(unknown.shape, dtype=torch.int32, device=unknown.device)
        pointnet2.three_nn_wrapper(unknown.float().contiguous(), _i32(unknown_batch_cnt).contiguous(),
                                   known.float().contiguous(), _i32(known_batch_cnt).contiguous(), dist2, idx)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """ref :264-302 — features (M,C), idx/weight (N,3) -> (N,C)."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor):
        assert tuple(idx.shape) == tuple(weight.shape) and idx.shape[1] == 3, "idx and weight must both be (N, 3)"
        ctx.three_interpolate_for_backward = (idx, weight, features.shape[0])
        output = features.new_zeros((idx.shape[0], features.shape[1]), dtype=torch.float32)
        pointnet2.three_interpolate_wrapper(features.float().contiguous(), idx.contiguous(), weight.float().contiguous(), output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, M = ctx.three_interpolate_for_backward
        grad_features = grad_out.new_zeros((M, grad_out.shape[1]), dtype=torch.float32)
        pointnet2.three_interpolate_grad_wrapper(grad_out.float().contiguous(), idx.contiguous(),
                                                 weight.float().contiguous(), grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class ThreeNNForVectorPoolByTwoStep(Function):
    """ref :305-353 — step 1 stacks every centre's neighbours within multiplier * max_neighbour_distance, step 2
    takes the three nearest of them for each of the centre's local grid-cell centres.
    -> (dist (M, G, 3) = sqrt(dist2), idx (M, G, 3) int32 GLOBAL or -1, tensor(avg_length_of_neighbor_idxs)).

    Upstream guesses the stack size from avg_length_of_neighbor_idxs and re-runs step 1 until it fits; here step 1
    is a count pass and a fill pass, so the stack is sized exactly after one host read of the total and
    avg_length_of_neighbor_idxs only matters as the value handed back (ceil(total / M), as upstream's last round)."""

    @staticmethod
    def forward(ctx, support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt,
                max_neighbour_distance, nsample, neighbor_type, avg_length_of_neighbor_idxs, num_total_grids,
                neighbor_distance_multiplier):
        num_new_xyz = new_xyz.shape[0]
        new_xyz_grid_centers = new_xyz_grid_centers.float().contiguous()
        new_xyz_grid_dist2 = new_xyz_grid_centers.new_zeros(new_xyz_grid_centers.shape)
        new_xyz_grid_idxs = torch.full(new_xyz_grid_centers.shape, -1, dtype=torch.int32, device=new_xyz.device)
        if num_new_xyz == 0:
            return new_xyz_grid_dist2, new_xyz_grid_idxs, torch.tensor(int(avg_length_of_neighbor_idxs))
        support_xyz, new_xyz = support_xyz.float().contiguous(), new_xyz.float().contiguous()
        xyz_batch_cnt, new_xyz_batch_cnt = _i32(xyz_batch_cnt).contiguous(), _i32(new_xyz_batch_cnt).contiguous()
        # same float product as upstream's python (:338) before it is narrowed to the wrapper's float argument
        distance = max_neighbour_distance * neighbor_distance_multiplier
        start_len = torch.zeros((num_new_xyz, 2), dtype=torch.int32, device=new_xyz.device)
        cumsum = torch.zeros((1,), dtype=torch.int32, device=new_xyz.device)
        pointnet2.local_neighbor_count(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, start_len, cumsum, distance,
                                       nsample, neighbor_type)
        total = int(cumsum.item())
        avg_length_of_neighbor_idxs = total // num_new_xyz + int(total % num_new_xyz > 0)
        stack_neighbor_idxs = torch.zeros((total,), dtype=torch.int32, device=new_xyz.device)
        pointnet2.local_neighbor_fill(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, stack_neighbor_idxs, start_len,
                                      distance, nsample, neighbor_type)
        pointnet2.query_three_nn_by_stacked_local_idxs_wrapper_stack(
            support_xyz, new_xyz, new_xyz_grid_centers, new_xyz_grid_idxs, new_xyz_grid_dist2,
            stack_neighbor_idxs, start_len, num_new_xyz, num_total_grids)
        return torch.sqrt(new_xyz_grid_dist2), new_xyz_grid_idxs, torch.tensor(avg_length_of_neighbor_idxs)

    @staticmethod
    def backward(ctx, *grads):
        return (None,) * 11


three_nn_for_vector_pool_by_two_step = ThreeNNForVectorPoolByTwoStep.apply


class VectorPoolWithVoxelQuery(Function):
    """ref :358-448 — every centre's neighbours fall into the cells of its local num_grid_x * y * z lattice; per cell the
    features (input channel i folded onto i % num_c_out_each_grid) and the local offsets are averaged (pooling_type 0)
    or taken from the first point (pooling_type 1).
    -> (new_features (M, G * num_c_out_each_grid), new_local_xyz (M, 3G), num_mean_points_per_grid (1,) int32 CPU,
        point_cnt_of_grid (M, G) int32); the gradient goes to support_features only.

    num_mean_points_per_grid is upstream's first guess for the size of the backward's index; it is not needed here
    (count pass, then an exactly sized fill) and only its final value, ceil(entries / M), is reproduced."""

    @staticmethod
    def forward(ctx, support_xyz: torch.Tensor, xyz_batch_cnt: torch.Tensor, support_features: torch.Tensor,
                new_xyz: torch.Tensor, new_xyz_batch_cnt: torch.Tensor, num_grid_x, num_grid_y, num_grid_z,
                max_neighbour_distance, num_c_out_each_grid, use_xyz,
                num_mean_points_per_grid=100, nsample=-1, neighbor_type=0, pooling_type=0):
        for name, t in (("support_xyz", support_xyz), ("support_features", support_features), ("xyz_batch_cnt", xyz_batch_cnt),
                        ("new_xyz", new_xyz), ("new_xyz_batch_cnt", new_xyz_batch_cnt)):
            assert t.is_contiguous(), f"{name} must be contiguous"
        num_total_grids = num_grid_x * num_grid_y * num_grid_z
        num_c_out = num_c_out_each_grid * num_total_grids
        N, num_c_in = support_features.shape
        M = new_xyz.shape[0]
        assert num_c_in % num_c_out_each_grid == 0, \
            f'the input channels ({num_c_in}) should be an integral multiple of num_c_out_each_grid({num_c_out_each_grid})'

        dev = new_xyz.device
        new_features = torch.zeros((M, num_c_out), dtype=torch.float32, device=dev)
        new_local_xyz = torch.zeros((M, 3 * num_total_grids), dtype=torch.float32, device=dev)
        point_cnt_of_grid = torch.zeros((M, num_total_grids), dtype=torch.int32, device=dev)
        grouped_idxs = torch.zeros((0, 3), dtype=torch.int32, device=dev)
        if M > 0:
            support_xyz, new_xyz = support_xyz.float(), new_xyz.float()
            xyz_batch_cnt, new_xyz_batch_cnt = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
            entry_start, total = pointnet2.vector_pool_count(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, num_grid_x,
                                                             num_grid_y, num_grid_z, max_neighbour_distance, nsample,
                                                             neighbor_type, pooling_type)
            num_cum_sum = int(total.item())
            num_mean_points_per_grid = num_cum_sum // M + int(num_cum_sum % M > 0)
            grouped_idxs = torch.zeros((num_cum_sum, 3), dtype=torch.int32, device=dev)
            pointnet2.vector_pool_fill(support_xyz, xyz_batch_cnt, support_features.float(), new_xyz, new_xyz_batch_cnt,
                                       new_features, new_local_xyz, point_cnt_of_grid, grouped_idxs, entry_start, num_grid_x,
                                       num_grid_y, num_grid_z, max_neighbour_distance, use_xyz, nsample, neighbor_type,
                                       pooling_type)

        normalizer = torch.clamp_min(point_cnt_of_grid[:, :, None].float(), min=1e-6)   # ref :416-420
        new_features = (new_features.view(-1, num_total_grids, num_c_out_each_grid) / normalizer).view(-1, num_c_out)
        if use_xyz:
            new_local_xyz = (new_local_xyz.view(-1, num_total_grids, 3) / normalizer).view(-1, num_total_grids * 3)

        num_mean_points_per_grid = torch.Tensor([num_mean_points_per_grid]).int()
        nsample = torch.Tensor([nsample]).int()
        ctx.vector_pool_for_backward = (point_cnt_of_grid, grouped_idxs, N, num_c_in)
        ctx.mark_non_differentiable(new_local_xyz, num_mean_points_per_grid, nsample, point_cnt_of_grid)
        return new_features, new_local_xyz, num_mean_points_per_grid, point_cnt_of_grid

    @staticmethod
    def backward(ctx, grad_new_features: torch.Tensor, grad_local_xyz: torch.Tensor, grad_num_cum_sum, grad_point_cnt_of_grid):
        point_cnt_of_grid, grouped_idxs, N, num_c_in = ctx.vector_pool_for_backward
        grad_support_features = grad_new_features.new_zeros((N, num_c_in), dtype=torch.float32)
        if grouped_idxs.shape[0] > 0:
            pointnet2.vector_pool_grad_wrapper(grad_new_features.float().contiguous(), point_cnt_of_grid, grouped_idxs,
                                               grad_support_features)
        return (None, None, grad_support_features) + (None,) * 12


vector_pool_with_voxel_query_op = VectorPoolWithVoxelQuery.apply
