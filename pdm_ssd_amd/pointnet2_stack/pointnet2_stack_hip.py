"""Drop-in for the part of the reference's native extension `pointnet2_stack_cuda` that the PointNet++ operators use
(ragged "stacked" batches).  Same function names, argument order and meaning as the pybind table at
/root/reference/pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:13-24:
    ball_query_wrapper, group_points_wrapper, group_points_grad_wrapper, three_nn_wrapper,
    three_interpolate_wrapper, three_interpolate_grad_wrapper, farthest_point_sampling_wrapper,
    stack_farthest_point_sampling_wrapper.
Not provided (outside this build): voxel_query_wrapper, the vector-pool functions.
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
