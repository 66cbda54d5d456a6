"""Drop-in for the part of the reference's native extension `pointnet2_stack_cuda` that the PointNet++ operators use
(ragged "stacked" batches).  Same function names, argument order and meaning as the pybind table at
/root/reference/pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:13-24:
    ball_query_wrapper, group_points_wrapper, group_points_grad_wrapper, three_nn_wrapper,
    three_interpolate_wrapper, three_interpolate_grad_wrapper, farthest_point_sampling_wrapper,
    stack_farthest_point_sampling_wrapper, voxel_query_wrapper (:14),
    query_stacked_local_neighbor_idxs_wrapper_stack, query_three_nn_by_stacked_local_idxs_wrapper_stack,
    vector_pool_wrapper, vector_pool_grad_wrapper (:25-30).
The vector-pool functions also come split into a count pass and a fill pass (local_neighbor_count / _fill,
vector_pool_count) so callers can size the stacked outputs exactly instead of re-running on overrun.
Tensors go to libpdmssd_hip.so as raw device pointers on the current torch stream; every argument is checked and
a Python exception raised on misuse or launch failure (the reference calls exit(-1)).
"""
import torch

from .. import _native
from ..pointnet2_batch.pointnet2_batch_hip import _check, _numel_at_least, _run, farthest_point_sampling_wrapper  # noqa: F401


def _cnt(name, t, B=None):
    _check(name, t, torch.int32)
    if t.dim() != 1 or (B is not None and t.numel() != B):
        raise ValueError(f"{name} must be a 1-D int32 tensor of the batch size")


def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
    _check("new_xyz", new_xyz, torch.float32); _check("xyz", xyz, torch.float32); _check("idx", idx, torch.int32)
    _cnt("new_xyz_batch_cnt", new_xyz_batch_cnt, B); _cnt("xyz_batch_cnt", xyz_batch_cnt, B)
    _numel_at_least("new_xyz", new_xyz, M * 3); _numel_at_least("idx", idx, M * nsample)
    _run("pdm_stack_ball_query", xyz, B, M, float(radius), nsample, new_xyz.data_ptr(), new_xyz_batch_cnt.data_ptr(),
         xyz.data_ptr(), xyz_batch_cnt.data_ptr(), idx.data_ptr())
    return 1


def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    _check("features", features, torch.float32); _check("idx", idx, torch.int32); _check("out", out, torch.float32)
    _cnt("features_batch_cnt", features_batch_cnt, B); _cnt("idx_batch_cnt", idx_batch_cnt, B)
    _numel_at_least("idx", idx, M * nsample); _numel_at_least("out", out, M * C * nsample)
    _run("pdm_stack_group_points", features, B, M, C, nsample, features.data_ptr(), features_batch_cnt.data_ptr(),
         idx.data_ptr(), idx_batch_cnt.data_ptr(), out.data_ptr())
    return 1


def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
    _check("grad_out", grad_out, torch.float32); _check("idx", idx, torch.int32)
    _check("grad_features", grad_features, torch.float32)
    _cnt("features_batch_cnt", features_batch_cnt, B); _cnt("idx_batch_cnt", idx_batch_cnt, B)
    _numel_at_least("grad_out", grad_out, M * C * nsample); _numel_at_least("grad_features", grad_features, N * C)
    _run("pdm_stack_group_points_grad", grad_out, B, M, C, N, nsample, grad_out.data_ptr(), idx.data_ptr(),
         idx_batch_cnt.data_ptr(), features_batch_cnt.data_ptr(), grad_features.data_ptr())
    return 1


def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
    _check("unknown", unknown, torch.float32); _check("known", known, torch.float32)
    _check("dist2", dist2, torch.float32); _check("idx", idx, torch.int32)
    B = unknown_batch_cnt.numel()
    _cnt("unknown_batch_cnt", unknown_batch_cnt, B); _cnt("known_batch_cnt", known_batch_cnt, B)
    N = unknown.shape[0]
    _numel_at_least("dist2", dist2, N * 3); _numel_at_least("idx", idx, N * 3)
    _run("pdm_stack_three_nn", unknown, B, N, unknown.data_ptr(), unknown_batch_cnt.data_ptr(), known.data_ptr(),
         known_batch_cnt.data_ptr(), dist2.data_ptr(), idx.data_ptr())


def three_interpolate_wrapper(features, idx, weight, out):
    _check("features", features, torch.float32); _check("idx", idx, torch.int32)
    _check("weight", weight, torch.float32); _check("out", out, torch.float32)
    N, C = idx.shape[0], features.shape[1]
    _numel_at_least("weight", weight, N * 3); _numel_at_least("out", out, N * C)
    _run("pdm_stack_three_interpolate", features, N, C, features.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr())


def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
    _check("grad_out", grad_out, torch.float32); _check("idx", idx, torch.int32)
    _check("weight", weight, torch.float32); _check("grad_features", grad_features, torch.float32)
    N, C = grad_out.shape
    _numel_at_least("idx", idx, N * 3); _numel_at_least("weight", weight, N * 3)
    _run("pdm_stack_three_interpolate_grad", grad_out, N, C, grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(),
         grad_features.data_ptr())


def stack_farthest_point_sampling_wrapper(xyz, temp, xyz_batch_cnt, idx, num_sampled_points):
    _check("xyz", xyz, torch.float32); _check("temp", temp, torch.float32); _check("idx", idx, torch.int32)
    B = xyz_batch_cnt.numel()
    _cnt("xyz_batch_cnt", xyz_batch_cnt, B); _cnt("num_sampled_points", num_sampled_points, B)
    _numel_at_least("temp", temp, xyz.shape[0])
    # the largest per-sample count picks the register-resident instantiation (one host sync, like the reference's
    # npoint.sum().item() in StackFarthestPointSampling.forward)
    counts = torch.stack([xyz_batch_cnt, num_sampled_points]).cpu()
    if int(counts[0].sum()) != xyz.shape[0]:
        raise ValueError("xyz_batch_cnt does not sum to the number of points")
    if bool(((counts[1] > 0) & (counts[0] < 1)).any()):
        raise ValueError("a sample with no points cannot be sampled")
    _numel_at_least("idx", idx, int(counts[1].sum()))
    _run("pdm_stack_furthest_point_sampling", xyz, B, int(counts[0].max()) if B else 0, xyz.data_ptr(), temp.data_ptr(),
         xyz_batch_cnt.data_ptr(), idx.data_ptr(), num_sampled_points.data_ptr())
    return 1


# ---- voxel query and vector pool (pointnet2_api.cpp:14, :25-30) -------------------------------------------------

def voxel_query_wrapper(M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, point_indices, idx):
    """voxel_query.cpp:25 — idx (M, nsample) caller-zeroed; -1 in slot 0 of a row that found nothing."""
    _check("new_xyz", new_xyz, torch.float32); _check("xyz", xyz, torch.float32)
    _check("new_coords", new_coords, torch.int32); _check("point_indices", point_indices, torch.int32)
    _check("idx", idx, torch.int32)
    _numel_at_least("new_xyz", new_xyz, M * 3); _numel_at_least("new_coords", new_coords, M * 4)
    _numel_at_least("idx", idx, M * nsample)
    if point_indices.dim() != 4 or tuple(point_indices.shape[1:]) != (R1, R2, R3):
        raise ValueError(f"point_indices must be (B, {R1}, {R2}, {R3}), got {tuple(point_indices.shape)}")
    _run("pdm_stack_voxel_query", xyz, M, R1, R2, R3, nsample, float(radius), z_range, y_range, x_range, new_xyz.data_ptr(),
         xyz.data_ptr(), new_coords.data_ptr(), point_indices.data_ptr(), idx.data_ptr())
    return 1


def _local_query_checks(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, start_len):
    _check("support_xyz", support_xyz, torch.float32); _check("new_xyz", new_xyz, torch.float32)
    B = xyz_batch_cnt.numel()
    _cnt("xyz_batch_cnt", xyz_batch_cnt, B); _cnt("new_xyz_batch_cnt", new_xyz_batch_cnt, B)
    _check("start_len", start_len, torch.int32)
    M = new_xyz.shape[0]
    _numel_at_least("start_len", start_len, M * 2)
    return B, M


def local_neighbor_count(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, start_len, cumsum, max_neighbour_distance,
                         nsample, neighbor_type):
    """Pass 1: start_len (M,2) = [exclusive prefix + cumsum, length]; cumsum[0] += total.  No stack is touched."""
    B, M = _local_query_checks(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, start_len)
    _check("cumsum", cumsum, torch.int32)
    _run("pdm_stack_local_neighbor_count", new_xyz, support_xyz.data_ptr(), xyz_batch_cnt.data_ptr(), new_xyz.data_ptr(),
         new_xyz_batch_cnt.data_ptr(), start_len.data_ptr(), cumsum.data_ptr(), float(max_neighbour_distance), B, M, nsample,
         neighbor_type)


def local_neighbor_fill(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, stack_neighbor_idxs, start_len,
                        max_neighbour_distance, nsample, neighbor_type):
    """Pass 2: the lists at start_len's offsets, cut at the length of stack_neighbor_idxs."""
    B, M = _local_query_checks(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, start_len)
    _check("stack_neighbor_idxs", stack_neighbor_idxs, torch.int32)
    _run("pdm_stack_local_neighbor_fill", new_xyz, support_xyz.data_ptr(), xyz_batch_cnt.data_ptr(), new_xyz.data_ptr(),
         new_xyz_batch_cnt.data_ptr(), stack_neighbor_idxs.data_ptr(), start_len.data_ptr(), stack_neighbor_idxs.numel(),
         float(max_neighbour_distance), B, M, nsample, neighbor_type)


def query_stacked_local_neighbor_idxs_wrapper_stack(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, stack_neighbor_idxs,
                                                    start_len, cumsum, avg_length_of_neighbor_idxs, max_neighbour_distance,
                                                    nsample, neighbor_type):
    """vector_pool.cpp:32 — one call, stack of avg_length * M slots; cumsum[0] > that capacity tells the caller to retry."""
    B, M = _local_query_checks(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, start_len)
    _check("stack_neighbor_idxs", stack_neighbor_idxs, torch.int32); _check("cumsum", cumsum, torch.int32)
    _numel_at_least("stack_neighbor_idxs", stack_neighbor_idxs, avg_length_of_neighbor_idxs * M)
    _run("pdm_stack_query_local_neighbor_idxs", new_xyz, support_xyz.data_ptr(), xyz_batch_cnt.data_ptr(), new_xyz.data_ptr(),
         new_xyz_batch_cnt.data_ptr(), stack_neighbor_idxs.data_ptr(), start_len.data_ptr(), cumsum.data_ptr(),
         int(avg_length_of_neighbor_idxs), float(max_neighbour_distance), B, M, nsample, neighbor_type)
    return 0


def query_three_nn_by_stacked_local_idxs_wrapper_stack(support_xyz, new_xyz, new_xyz_grid_centers, new_xyz_grid_idxs,
                                                       new_xyz_grid_dist2, stack_neighbor_idxs, start_len, M, num_total_grids):
    """vector_pool.cpp:75 — dist2 / idxs (M, num_total_grids, 3)."""
    _check("support_xyz", support_xyz, torch.float32); _check("new_xyz_grid_centers", new_xyz_grid_centers, torch.float32)
    _check("new_xyz_grid_idxs", new_xyz_grid_idxs, torch.int32); _check("new_xyz_grid_dist2", new_xyz_grid_dist2, torch.float32)
    _check("stack_neighbor_idxs", stack_neighbor_idxs, torch.int32); _check("start_len", start_len, torch.int32)
    for name, t in (("new_xyz_grid_centers", new_xyz_grid_centers), ("new_xyz_grid_idxs", new_xyz_grid_idxs),
                    ("new_xyz_grid_dist2", new_xyz_grid_dist2)):
        _numel_at_least(name, t, M * num_total_grids * 3)
    _numel_at_least("start_len", start_len, M * 2)
    _run("pdm_stack_three_nn_by_local_idxs", new_xyz_grid_centers, support_xyz.data_ptr(), new_xyz_grid_centers.data_ptr(),
         new_xyz_grid_idxs.data_ptr(), new_xyz_grid_dist2.data_ptr(), stack_neighbor_idxs.data_ptr(), start_len.data_ptr(),
         stack_neighbor_idxs.numel(), M, num_total_grids)
    return 0


def _pool_checks(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    _check("support_xyz", support_xyz, torch.float32); _check("new_xyz", new_xyz, torch.float32)
    B = xyz_batch_cnt.numel()
    _cnt("xyz_batch_cnt", xyz_batch_cnt, B); _cnt("new_xyz_batch_cnt", new_xyz_batch_cnt, B)
    return B, new_xyz.shape[0]


def vector_pool_count(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, num_grid_x, num_grid_y, num_grid_z,
                      max_neighbour_distance, nsample, neighbor_type, pooling_type):
    """Pass 1 of the pooling: -> (entry_start (M,) int32, total (1,) int32 on the device)."""
    B, M = _pool_checks(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
    work = torch.zeros((2 * M + 1,), dtype=torch.int32, device=new_xyz.device)
    entry_start, entry_cnt, total = work[:M], work[M:2 * M], work[2 * M:]
    _run("pdm_stack_vector_pool_count", new_xyz, support_xyz.data_ptr(), xyz_batch_cnt.data_ptr(), new_xyz.data_ptr(),
         new_xyz_batch_cnt.data_ptr(), entry_start.data_ptr(), entry_cnt.data_ptr(), total.data_ptr(), num_grid_x, num_grid_y,
         num_grid_z, float(max_neighbour_distance), B, M, nsample, neighbor_type, pooling_type)
    return entry_start, total


def vector_pool_fill(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, new_features, new_local_xyz,
                     point_cnt_of_grid, grouped_idxs, entry_start, num_grid_x, num_grid_y, num_grid_z, max_neighbour_distance,
                     use_xyz, nsample, neighbor_type, pooling_type):
    """Pass 2 of the pooling: the sums, the counts and grouped_idxs (cut at its own length)."""
    B, M = _pool_checks(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
    _check("support_features", support_features, torch.float32); _check("new_features", new_features, torch.float32)
    _check("new_local_xyz", new_local_xyz, torch.float32); _check("point_cnt_of_grid", point_cnt_of_grid, torch.int32)
    _check("grouped_idxs", grouped_idxs, torch.int32); _check("entry_start", entry_start, torch.int32)
    G = num_grid_x * num_grid_y * num_grid_z
    if support_features.dim() != 2 or support_features.shape[0] != support_xyz.shape[0]:
        raise ValueError("support_features must be (N, C) with one row per support point")
    if new_features.dim() != 2 or new_features.shape[0] != M or new_features.shape[1] % G:
        raise ValueError(f"new_features must be (M, k * {G})")
    _numel_at_least("new_local_xyz", new_local_xyz, M * 3 * G); _numel_at_least("point_cnt_of_grid", point_cnt_of_grid, M * G)
    _numel_at_least("entry_start", entry_start, M)
    _run("pdm_stack_vector_pool", new_xyz, support_xyz.data_ptr(), support_features.data_ptr(), xyz_batch_cnt.data_ptr(),
         new_xyz.data_ptr(), new_features.data_ptr(), new_local_xyz.data_ptr(), new_xyz_batch_cnt.data_ptr(),
         point_cnt_of_grid.data_ptr(), grouped_idxs.data_ptr(), entry_start.data_ptr(), num_grid_x, num_grid_y, num_grid_z,
         float(max_neighbour_distance), B, M, support_features.shape[1], new_features.shape[1], int(bool(use_xyz)),
         grouped_idxs.shape[0], nsample, neighbor_type, pooling_type)


def vector_pool_wrapper(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, new_features, new_local_xyz,
                        point_cnt_of_grid, grouped_idxs, num_grid_x, num_grid_y, num_grid_z, max_neighbour_distance, use_xyz,
                        num_max_sum_points, nsample, neighbor_type, pooling_type):
    """vector_pool.cpp:113 — returns the number of grouped entries the call wants (one host read, as upstream's
    cudaMemcpy of the cursor); when that exceeds num_max_sum_points the outputs are left untouched and the caller is
    expected to come back with a larger grouped_idxs, exactly upstream's protocol."""
    entry_start, total = vector_pool_count(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, num_grid_x, num_grid_y,
                                           num_grid_z, max_neighbour_distance, nsample, neighbor_type, pooling_type)
    num_cum_sum = int(total.item())
    if num_cum_sum <= num_max_sum_points:
        _numel_at_least("grouped_idxs", grouped_idxs, num_max_sum_points * 3)
        vector_pool_fill(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, new_features, new_local_xyz,
                         point_cnt_of_grid, grouped_idxs, entry_start, num_grid_x, num_grid_y, num_grid_z,
                         max_neighbour_distance, use_xyz, nsample, neighbor_type, pooling_type)
    return num_cum_sum


def vector_pool_grad_wrapper(grad_new_features, point_cnt_of_grid, grouped_idxs, grad_support_features):
    """vector_pool.cpp:170 — grad_support_features (N, C_in) caller-zeroed."""
    _check("grad_new_features", grad_new_features, torch.float32); _check("point_cnt_of_grid", point_cnt_of_grid, torch.int32)
    _check("grouped_idxs", grouped_idxs, torch.int32); _check("grad_support_features", grad_support_features, torch.float32)
    M, num_c_out = grad_new_features.shape
    N, num_c_in = grad_support_features.shape
    G = point_cnt_of_grid.shape[1]
    if point_cnt_of_grid.shape[0] != M or num_c_out % G:
        raise ValueError("point_cnt_of_grid must be (M, G) with num_c_out a multiple of G")
    _run("pdm_stack_vector_pool_grad", grad_new_features, grad_new_features.data_ptr(), point_cnt_of_grid.data_ptr(),
         grouped_idxs.data_ptr(), grad_support_features.data_ptr(), N, M, num_c_out, num_c_in, G, grouped_idxs.shape[0])
    return 0
