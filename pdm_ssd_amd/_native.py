"""ctypes binding of libpdmssd_hip.so (the C ABI declared in include/pdmssd_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, an exception is
raised.  Nothing here imports the CPU oracle.
"""
import ctypes
import os

# torch first: its wheel bundles the HIP runtime (libamdhip64.so.7) this library must share with it —
# streams and device pointers handed across the C ABI are only meaningful inside ONE runtime instance.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpdmssd_hip.so")
ABI_VERSION = 1

_lib = None

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> argtypes (after the leading `void *stream`); mirrors include/pdmssd_hip.h
_SIGNATURES = {
    "pdm_ball_query": [_i, _i, _i, _f, _i, _vp, _vp, _vp],
    "pdm_ball_query_grid": [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_grid_build": [_i, _i, _f, _vp, _vp, ctypes.c_size_t],
    "pdm_ball_query_grid_prebuilt": [_i, _i, _i, _f, _i, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_three_nn_grid_prebuilt": [_i, _i, _i, _vp, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_three_nn_grid": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_group_points": [_i, _i, _i, _i, _i, _vp, _vp, _vp],
    "pdm_group_points_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp],
    "pdm_group_points_grad_ws": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_gather_points": [_i, _i, _i, _i, _vp, _vp, _vp],
    "pdm_gather_points_grad": [_i, _i, _i, _i, _vp, _vp, _vp],
    "pdm_furthest_point_sampling": [_i, _i, _i, _vp, _vp, _vp],
    "pdm_furthest_point_sampling_ws": [_i, _i, _i, _vp, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_furthest_point_sampling_status": [_i, _i, _vp, _vp],
    "pdm_furthest_point_sampling_jobs": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_topk_sampling": [_i, _i, _i, _vp, _vp],
    "pdm_three_nn": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pdm_three_interpolate": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdm_three_interpolate_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdm_three_interpolate_grad_ws": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_query_and_group": [_i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp],
    "pdm_group_concat": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdm_group_concat_cl": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i],
    "pdm_group_concat_cl_grad": [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_group_concat_cl_ld": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i],
    "pdm_group_concat_cl_ld_f": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i],
    "pdm_group_concat_cl_grad_ld": [_i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_sa_mlp_fused": [_i] * 5 + [_vp] * 4 + [_i, _vp, _vp, _vp, _vp, _i, _i, _i],
    "pdm_fp_mlp_fused": [_i] * 5 + [_vp] * 4 + [_i, _vp, _vp, _vp, _vp, _i, _i],
    "pdm_copy_many": [_i, _vp, _vp, _vp],
    "pdm_copy_many_dyn": [_i, _vp, _vp, _vp, _vp, _vp],
    "pdm_mark_time": [_vp],
    "pdm_boxes_overlap_bev": [_i, _vp, _i, _vp, _vp],
    "pdm_boxes_iou_bev": [_i, _vp, _i, _vp, _vp],
    "pdm_boxes_aligned_overlap_bev": [_i, _vp, _vp, _vp],
    "pdm_nms": [_i, _vp, _f, _i, _vp, ctypes.c_size_t, _vp, _vp],
    "pdm_points_in_boxes": [_i, _i, _i, _vp, _vp, _vp],
    "pdm_bev_depthwise3x3": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i],
    "pdm_bev_depthwise3x3_wgrad": [_i, _i, _i, _i, _vp, _vp, _vp],
    "pdm_bev_depthwise3x3_t": [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i],
    "pdm_bev_depthwise3x3_wgrad_t": [_i, _i, _i, _i, _vp, _vp, _i, _vp],
    "pdm_point_head_decode": [ctypes.c_longlong, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp],
    "pdm_bev_head_fused": [_i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i],
    "pdm_sample_points": [_i, _i, ctypes.c_uint, _i, _vp, _vp, _vp, _vp],
    "pdm_stack_ball_query": [_i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp],
    "pdm_stack_group_points": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdm_stack_group_points_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdm_stack_three_nn": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdm_stack_three_interpolate": [_i, _i, _vp, _vp, _vp, _vp],
    "pdm_stack_three_interpolate_grad": [_i, _i, _vp, _vp, _vp, _vp],
    "pdm_stack_furthest_point_sampling": [_i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdm_bn_relu_pool_forward": [_i, ctypes.c_longlong, _i, _i] + [_vp] * 8 + [_f, _f, _vp, _vp, _vp, _vp, _i],
    "pdm_bn_relu_pool_backward": [_i, ctypes.c_longlong, _i, _i] + [_vp] * 10 + [_i],
    "pdm_bn_relu_forward": [_i, _i, ctypes.c_longlong, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i],
    "pdm_bn_relu_forward_stats": [_i, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i, _i],
    "pdm_bn_relu_backward": [_i, _i, ctypes.c_longlong, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _i],
    "pdm_bn_relu_backward_stats": [_i, _i, ctypes.c_longlong, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _i],
    "pdm_bn_relu_backward_apply": [_i, _i, ctypes.c_longlong, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _i],
    "pdm_stack_voxel_query": [_i, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdm_stack_local_neighbor_count": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i],
    "pdm_stack_local_neighbor_fill": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _f, _i, _i, _i, _i],
    "pdm_stack_query_local_neighbor_idxs": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _i, _i, _i],
    "pdm_stack_three_nn_by_local_idxs": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i],
    "pdm_stack_vector_pool_count": [_vp] * 7 + [_i, _i, _i, _f, _i, _i, _i, _i, _i],
    "pdm_stack_vector_pool": [_vp] * 10 + [_i, _i, _i, _f] + [_i] * 9,
    "pdm_stack_vector_pool_grad": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "pdm_three_nn_weights": [ctypes.c_longlong, _vp, _vp, _vp],
    "pdm_rows_mlp_fused": [_i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i],
    "pdm_point_head_loss": [ctypes.c_longlong, _i, _i, _i, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong,
                            _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_heatmap_targets": [_i, _i, _i, _i, _i, _vp, _f, _f, _f, _f, _f, ctypes.c_double, _i, _i, _vp],
    "pdm_heatmap_focal_loss": [_i, _i, _i, _i, _vp, _i, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _vp, _f, _vp, _vp,
                               _vp, ctypes.c_size_t],
    "pdm_rows_mlp_x3": [_i, _i, _vp, _i, _vp, _vp, ctypes.c_size_t, _vp, _i, _vp, _i, _i],
    "pdm_rows_mlp_fused_pair": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _i, _i],
    "pdm_fp_head_fused": [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i,
                          _vp, _i, _i],
    "pdm_sa_mlp_fused_pre": [_i] * 4 + [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i],
    "pdm_fp_mlp_fused_pre": [_i] * 4 + [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i],
    "pdm_sa_pack": [_i, _i, _i, _i, _vp, _vp, ctypes.c_size_t, _vp, _vp],
    "pdm_sa_pack_pair": [_i, _i, _i, _vp, _vp, _vp, ctypes.c_size_t, _vp, _vp],
    "pdm_sa_mlp_packed_pair": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "pdm_sa_mlp_packed": [_i] * 5 + [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i],
    "pdm_interp_concat_rows": [_i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp],
    "pdm_interp_concat_rows_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_interp_concat_rows_grad_out": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, ctypes.c_size_t],
    "pdm_tg_gemm_nt": [ctypes.c_longlong, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _vp, _vp],
    "pdm_tg_gemm_nt_dy": [ctypes.c_longlong, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong,
                          _vp, ctypes.c_longlong, _vp, _vp],
    "pdm_tg_gemm_nt_bs": [ctypes.c_longlong, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp,
                          ctypes.c_longlong, _vp, _vp],
    "pdm_tg_gemm_nt_dy_bs": [ctypes.c_longlong, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp,
                             ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _vp, _vp, ctypes.c_longlong, _vp, _vp],
    "pdm_bn_finalize_bwd_stats": [ctypes.c_longlong, _i, _vp, _vp, _vp, _i],
    "pdm_tg_gemm_nt_pool": [ctypes.c_longlong, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _vp, _vp,
                            _i, _vp, _vp, _vp, _vp],
    "pdm_bn_relu_pool_forward_kept": [_i, ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i, _i],
    "pdm_tg_wgrad": [ctypes.c_longlong, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _i, _vp, ctypes.c_size_t, _vp],
    "pdm_bn_forward_coef": [_i, ctypes.c_longlong, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp],
    "pdm_bn_finalize_stats": [ctypes.c_longlong, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i],
    "pdm_tg_colsum": [ctypes.c_longlong, _i, _vp, ctypes.c_longlong, _vp, _vp],
    "pdm_tg_pack_weight": [_i, _i, _vp, _vp, _i, _vp, _i],
    "pdm_tg_pack_weight_pair": [_i, _i, _vp, _vp, _vp, _i, _i],
    "pdm_tg_pack_weight_many": [_i, _vp, ctypes.c_longlong],
    "pdm_tune_fps_variant": None,
    "pdm_tune_fused_waves": None,
    "pdm_tune_fused_tiles": None,
    "pdm_tune_fused_groups": None,
    "pdm_tune_fused_wg_per_cu": None,
    "pdm_tune_fused_lds_cap": None,
    "pdm_tune_fused_reg": None,
    "pdm_tune_bq_quad": None,
    "pdm_tune_bq_heavy": None,
    "pdm_tune_bq_cpw": None,
    "pdm_tune_copy_variant": None,
    "pdm_tune_group_nt": None,
    "pdm_tune_group_lds_floor": None,
    "pdm_tune_copy_max_wg": None,
    "pdm_tune_bq_small_waves": None,
    "pdm_tune_bq_dense_ppc": None,
    "pdm_tune_grid_split": None,
    "pdm_tune_group_rows": None,
    "pdm_tune_fused_gemm": None,
    "pdm_tune_fused_chain": None,
    "pdm_tune_fused_pair": None,
    "pdm_tune_sa_pair": None,
    "pdm_tune_rows_x3_wg_per_cu": None,
    "pdm_tune_fp_chain_pad_lds": None,
    "pdm_tune_fp_chain_nt": None,
    "pdm_tune_fp_chain_mask": None,
    "pdm_tune_rows_chain_wg_per_cu": None,
    "pdm_tune_rows_chain_dw_wg_per_cu": None,
    "pdm_tune_fp_head_tiles": None,
    "pdm_tune_rows_chain_xcd": None,
    "pdm_tune_fused_swz": None,
    "pdm_scatter_bev": [_i, _i, _i, _i, _vp, _vp, _vp, _vp] + [_f] * 9 + [_i] * 7 + [_vp, _vp],
    "pdm_gather_bev": [_i, _i, _i, _i, _vp, _vp, _vp, _vp] + [_f] * 9 + [_i] * 6 + [_i, _f, _vp, _vp, _vp, ctypes.c_size_t],
    "pdm_bev_normalize_grad": [_i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp],
    "pdm_bev_normalize": [_i, _i, _i, _i, _i, _i, _f, _vp, _vp],
    "pdm_scatter_bev_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp] + [_f] * 9 + [_i] * 7 + [_vp] * 5,
    "pdm_scatter_bev_grad_normalized": [_i, _i, _i, _i, _vp, _vp, _vp, _vp] + [_f] * 9 + [_i] * 7 + [_vp, _vp, _f] + [_vp] * 4,
}
EXPORTS = ["pdm_abi_version", "pdm_last_error", "pdm_ball_query_grid_workspace_bytes",
           "pdm_three_nn_grid_workspace_bytes", "pdm_furthest_point_sampling_ws_bytes",
           "pdm_fps_max_coresident_workgroups",
           "pdm_gather_bev_workspace_bytes", "pdm_nms_workspace_bytes", "pdm_sa_pack_workspace_bytes",
           "pdm_sa_pack_rows", "pdm_rows_mlp_x3_stream_bytes", "pdm_point_head_loss_workspace_bytes", "pdm_heatmap_focal_loss_workspace_bytes", "pdm_three_interpolate_grad_ws_bytes",
           "pdm_group_points_grad_ws_bytes", "pdm_group_concat_cl_grad_ws_bytes", "pdm_bn_parts", "pdm_bn_pool_parts",
           "pdm_tg_stats_parts", "pdm_tg_dy_stats_parts", "pdm_tg_wgrad_ws_bytes", "pdm_tg_colsum_ws_floats"] + list(_SIGNATURES)


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load libpdmssd_hip.so once; raise loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} not found: build it with `make -C pdm_ssd_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "pdm_ssd_amd has no CPU or PyTorch fallback for its operators.")
        l = ctypes.CDLL(LIB_PATH)
        l.pdm_abi_version.restype = _i
        l.pdm_last_error.restype = ctypes.c_char_p
        l.pdm_ball_query_grid_workspace_bytes.restype = ctypes.c_size_t
        l.pdm_ball_query_grid_workspace_bytes.argtypes = [_i, _i]
        l.pdm_furthest_point_sampling_ws_bytes.restype = ctypes.c_size_t
        l.pdm_furthest_point_sampling_ws_bytes.argtypes = [_i, _i]
        l.pdm_fps_max_coresident_workgroups.restype = _i
        l.pdm_fps_max_coresident_workgroups.argtypes = []
        l.pdm_gather_bev_workspace_bytes.restype = ctypes.c_size_t
        l.pdm_gather_bev_workspace_bytes.argtypes = [_i] * 6
        l.pdm_three_nn_grid_workspace_bytes.restype = ctypes.c_size_t
        l.pdm_three_nn_grid_workspace_bytes.argtypes = [_i, _i]
        l.pdm_nms_workspace_bytes.restype = ctypes.c_size_t
        l.pdm_sa_pack_workspace_bytes.restype = ctypes.c_size_t
        l.pdm_sa_pack_workspace_bytes.argtypes = [_i, _i]
        l.pdm_group_concat_cl_grad_ws_bytes.restype = ctypes.c_size_t
        l.pdm_group_concat_cl_grad_ws_bytes.argtypes = [_i, _i, _i, _i]
        l.pdm_group_points_grad_ws_bytes.restype = ctypes.c_size_t
        l.pdm_group_points_grad_ws_bytes.argtypes = [_i, _i, _i, _i]
        l.pdm_three_interpolate_grad_ws_bytes.restype = ctypes.c_size_t
        l.pdm_three_interpolate_grad_ws_bytes.argtypes = [_i, _i, _i]
        l.pdm_point_head_loss_workspace_bytes.restype = ctypes.c_size_t
        l.pdm_point_head_loss_workspace_bytes.argtypes = [ctypes.c_longlong]
        l.pdm_heatmap_focal_loss_workspace_bytes.restype = ctypes.c_size_t
        l.pdm_heatmap_focal_loss_workspace_bytes.argtypes = [ctypes.c_longlong]
        l.pdm_rows_mlp_x3_stream_bytes.restype = ctypes.c_size_t
        l.pdm_rows_mlp_x3_stream_bytes.argtypes = [_i, _vp]
        l.pdm_sa_pack_rows.restype = ctypes.c_size_t
        l.pdm_sa_pack_rows.argtypes = [_i, _i, _i]
        l.pdm_nms_workspace_bytes.argtypes = [_i]
        l.pdm_bn_pool_parts.restype = _i
        l.pdm_bn_pool_parts.argtypes = [_i, ctypes.c_longlong, _i]
        l.pdm_tg_colsum_ws_floats.restype = ctypes.c_size_t
        l.pdm_tg_colsum_ws_floats.argtypes = [ctypes.c_longlong, _i]
        l.pdm_tg_stats_parts.restype = _i
        l.pdm_tg_stats_parts.argtypes = [ctypes.c_longlong, _i]
        l.pdm_tg_dy_stats_parts.restype = _i
        l.pdm_tg_dy_stats_parts.argtypes = [ctypes.c_longlong, _i]
        l.pdm_tg_wgrad_ws_bytes.restype = ctypes.c_size_t
        l.pdm_tg_wgrad_ws_bytes.argtypes = [ctypes.c_longlong, _i, _i]
        l.pdm_bn_parts.restype = _i
        l.pdm_bn_parts.argtypes = [_i, ctypes.c_longlong, _i, ctypes.c_longlong]
        if l.pdm_abi_version() != ABI_VERSION:
            raise NativeLibraryError(
                f"libpdmssd_hip.so ABI {l.pdm_abi_version()} != expected {ABI_VERSION}; rebuild it")
        for name, args in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = _i
            fn.argtypes = [_i] if args is None else [_vp] + args
        _lib = l
    return _lib


def call(name, stream, *args):
    """Invoke an entry point on `stream` (an int hipStream_t); raise on a non-zero return."""
    l = lib()
    rc = getattr(l, name)(stream, *args)
    if rc != 0:
        msg = l.pdm_last_error().decode("utf-8", "replace")
        raise NativeLibraryError(f"{name} failed with code {rc}: {msg}")


# Cooperating-workgroup FPS calls (n > 16384) whose status word has not been read yet: (workspace, b, n, event).
# The word is read without stalling the caller: when a later call finds the event complete, or in fps_check().
_fps_pending = []


def fps_watch(ws, b, n):
    """Remember a cooperating-workgroup FPS call for a deferred status check (not during graph capture: a captured
    launch is checked by whoever replays the graph, with fps_check_workspace)."""
    import torch
    if torch.cuda.is_current_stream_capturing():
        return
    fps_check(wait=False)
    ev = torch.cuda.Event()
    ev.record()
    _fps_pending.append((ws, b, n, ev))


def fps_check_workspace(ws, b, n):
    """Synchronise and raise if the cooperating-workgroup FPS that used `ws` gave up waiting for a peer workgroup."""
    import torch
    flag = ctypes.c_int(0)
    call("pdm_furthest_point_sampling_status", torch.cuda.current_stream(ws.device).cuda_stream, b, n, ws.data_ptr(),
         ctypes.cast(ctypes.pointer(flag), ctypes.c_void_p))
    if flag.value:
        raise NativeLibraryError(
            f"furthest_point_sampling ({b} clouds x {n} points): a workgroup gave up waiting for its peers — they were "
            "not co-resident (device shared, partitioned or CU-masked); the sample indices of that call are invalid")


def fps_check(wait=True):
    """Check the status words of earlier cooperating-workgroup FPS calls: all of them (synchronising) or, with
    wait=False, those that have finished."""
    keep = []
    for item in _fps_pending:
        ws, b, n, ev = item
        if wait or ev.query():
            fps_check_workspace(ws, b, n)
        else:
            keep.append(item)
    _fps_pending[:] = keep


def copy_many(dst, src, live=None):
    """dst[k].copy_(src[k]) for lists of same-shaped contiguous CUDA tensors, in one kernel launch.
    live[k] = None | (count, unit): `count` a 1-element int32 CUDA tensor (view) read ON THE DEVICE when the copy runs —
    only the first count * unit bytes of buffer k are live and copied (worst-case-sized buffers, fused.sa_pack)."""
    import torch
    n = len(dst)
    assert n == len(src)
    if n == 0:
        return
    if live is not None and any(x is not None for x in live):
        assert len(live) == n
        for d, s_ in zip(dst, src):
            assert d.is_contiguous() and s_.is_contiguous() and d.dtype == s_.dtype and d.shape == s_.shape, (d.shape, s_.shape)
        P = ctypes.c_void_p * n
        call("pdm_copy_many_dyn", torch.cuda.current_stream(dst[0].device).cuda_stream, n,
             ctypes.cast(P(*[d.data_ptr() for d in dst]), ctypes.c_void_p),
             ctypes.cast(P(*[s_.data_ptr() for s_ in src]), ctypes.c_void_p),
             ctypes.cast((ctypes.c_size_t * n)(*[d.numel() * d.element_size() for d in dst]), ctypes.c_void_p),
             ctypes.cast(P(*[None if x is None else x[0].data_ptr() for x in live]), ctypes.c_void_p),
             ctypes.cast((ctypes.c_uint * n)(*[0 if x is None else int(x[1]) for x in live]), ctypes.c_void_p))
        return
    for d, s_ in zip(dst, src):
        assert d.is_contiguous() and s_.is_contiguous() and d.dtype == s_.dtype and d.shape == s_.shape, (d.shape, s_.shape)
    P = ctypes.c_void_p * n
    dp = P(*[d.data_ptr() for d in dst])
    sp = P(*[s_.data_ptr() for s_ in src])
    nb = (ctypes.c_size_t * n)(*[d.numel() * d.element_size() for d in dst])
    call("pdm_copy_many", torch.cuda.current_stream(dst[0].device).cuda_stream, n,
         ctypes.cast(dp, ctypes.c_void_p), ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(nb, ctypes.c_void_p))
