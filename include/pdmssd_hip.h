/*
 * include/pdmssd_hip.h — C ABI of libpdmssd_hip.so (hand-written HIP kernels for gfx950 / MI355X).
 *
 * This is the drop-in boundary for PDM-SSD's point-cloud hot path.  Each entry point replaces one
 * function of the reference's native extension `pointnet2_batch_cuda`
 * (/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:10-24); the
 * reference-side line each one replaces is cited on the declaration.  Integer/float arguments keep
 * the reference's names, order and meaning; tensors arrive as raw device pointers.
 *
 * Contract (differs from the reference only where the reference is unsafe):
 *   - `stream` is a hipStream_t (NULL = the null stream).  The reference launches on the legacy
 *     default stream with no device guard; here the caller picks the stream and the device is the
 *     one current on the calling thread.  All calls are asynchronous, re-entrant and stateless.
 *   - Ownership: the CALLER allocates and pre-initialises every output / scratch buffer exactly as
 *     the reference's Python does (idx zero-filled for ball_query, temp filled with 1e10 for FPS,
 *     grad buffers zero-filled); the callee only writes through the pointers it is given and
 *     retains nothing.
 *   - Errors: return 0 on success; a positive value is a hipError_t from the launch, a negative
 *     value an argument error (PDM_E_*).  Nothing here ever calls exit() (the reference does:
 *     ball_query_gpu.cu:68-72).  pdm_last_error() returns a thread-local message.
 *   - All tensors are contiguous; data is fp32, indices are int32; 64-bit offsets internally.
 *   - No host synchronisation, allocation or memcpy inside any call: every entry point is
 *     hipGraph-capturable.
 */
#ifndef PDMSSD_HIP_H
#define PDMSSD_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDM_ABI_VERSION 1

#define PDM_E_BADARG (-1)   /* negative size, null pointer, unsupported parameter */
#define PDM_E_TOOLARGE (-2) /* size exceeds what the kernel's grid/indexing supports */

int pdm_abi_version(void);
const char *pdm_last_error(void);

/* ---- pointnet2_batch operators ------------------------------------------------------------ */

/* replaces ball_query_wrapper_fast            ball_query.cpp:29-39 -> ball_query_gpu.cu:15-73
 * new_xyz (B,M,3), xyz (B,N,3) -> idx (B,M,nsample), caller-zeroed. */
int pdm_ball_query(void *stream, int b, int n, int m, float radius, int nsample,
                   const float *new_xyz, const float *xyz, int *idx);

/* replaces group_points_wrapper_fast          group_points.cpp:27-36 -> group_points_gpu.cu:53-92
 * points (B,C,N), idx (B,npoints,nsample) -> out (B,C,npoints,nsample). */
int pdm_group_points(void *stream, int b, int c, int n, int npoints, int nsample,
                     const float *points, const int *idx, float *out);

/* replaces group_points_grad_wrapper_fast     group_points.cpp:15-24 -> group_points_gpu.cu:14-50
 * grad_out (B,C,npoints,nsample), idx -> grad_points (B,C,N), caller-zeroed, accumulated. */
int pdm_group_points_grad(void *stream, int b, int c, int n, int npoints, int nsample,
                          const float *grad_out, const int *idx, float *grad_points);

/* QueryAndGroup's concat (pointnet2_utils.py:249-257) in channels-last memory for the training path: out (B, M, nsample,
 * 3+C) = an NHWC view of the reference's (B, 3+C, M, nsample) tensor, fp32 or (out_bf16 = 1) bf16 rounded to nearest even
 * from the fp32 value; feat_pm (B, N, C) point-major; idx (B, M, nsample) from pdm_ball_query. */
int pdm_group_concat_cl(void *stream, int b, int n, int m, int c, int nsample, const float *xyz, const float *new_xyz,
                        const float *feat_pm, const int *idx, void *out, int out_bf16);
/* Its backward: grad (B, M, nsample, 3+C) fp32 or bf16 -> grad_feat_pm (B, N, C) fp32, fully written (the xyz channels carry
 * no gradient, as in the reference graph).  Scatter inverted into CSR lists in `workspace`
 * (pdm_group_concat_cl_grad_ws_bytes bytes), then one wave per source point adds its rows up: no atomics.  n <= 16384. */
size_t pdm_group_concat_cl_grad_ws_bytes(int b, int n, int m, int nsample);
int pdm_group_concat_cl_grad(void *stream, int b, int n, int m, int c, int nsample, const void *grad, int grad_bf16,
                             const int *idx, float *grad_feat_pm, void *workspace, size_t workspace_bytes);
/* Both with a row stride ld >= 3 + C: out / grad are (B, M, nsample, ld); the forward writes the channels 3 + C .. ld - 1 as
 * ZEROS (padding to 16-byte rows for the bf16 contractions of csrc/train_gemm.hip, ld = 3 + C rounded up to 8). */
int pdm_group_concat_cl_ld(void *stream, int b, int n, int m, int c, int nsample, const float *xyz, const float *new_xyz,
                           const float *feat_pm, const int *idx, void *out, int out_bf16, int ld);
/* the same, source features fp32 (feat_bf16 = 0) or bf16 rows (1: with the bf16 result and ld % 8 == 0; a bf16 feature passes through exactly) */
int pdm_group_concat_cl_ld_f(void *stream, int b, int n, int m, int c, int nsample, const float *xyz, const float *new_xyz,
                             const void *feat_pm, int feat_bf16, const int *idx, void *out, int out_bf16, int ld);
int pdm_group_concat_cl_grad_ld(void *stream, int b, int n, int m, int c, int nsample, const void *grad, int grad_bf16, int ld,
                                const int *idx, float *grad_feat_pm, void *workspace, size_t workspace_bytes);

/* The same backward with a caller-provided workspace (pdm_group_points_grad_ws_bytes bytes): scatter inverted into CSR lists,
 * accumulated without atomics (as pdm_three_interpolate_grad_ws).  Forwards to the plain entry point when a grad_out row
 * (npoints * nsample floats) exceeds 128 KB or n > 16384. */
size_t pdm_group_points_grad_ws_bytes(int b, int npoints, int nsample, int n);
int pdm_group_points_grad_ws(void *stream, int b, int c, int n, int npoints, int nsample, const float *grad_out,
                             const int *idx, float *grad_points, void *workspace, size_t workspace_bytes);

/* replaces gather_points_wrapper_fast         sampling.cpp:14-22 -> sampling_gpu.cu:15-51
 * points (B,C,N), idx (B,npoints) -> out (B,C,npoints). */
int pdm_gather_points(void *stream, int b, int c, int n, int npoints, const float *points,
                      const int *idx, float *out);

/* replaces gather_points_grad_wrapper_fast    sampling.cpp:25-34 -> sampling_gpu.cu:53-90 */
int pdm_gather_points_grad(void *stream, int b, int c, int n, int npoints, const float *grad_out,
                           const int *idx, float *grad_points);

/* replaces farthest_point_sampling_wrapper    sampling.cpp:37-46 -> sampling_gpu.cu:100-260
 * points (B,N,3), temp (B,N) caller-filled with 1e10 (read as the initial min-distances and left
 * holding the final ones) -> idx (B,m).  Tie order identical to the reference's block reduction
 * for block_size = min(2^floor(log2 n), 1024). */
int pdm_furthest_point_sampling(void *stream, int b, int n, int m, const float *points,
                                float *temp, int *idx);

/* Same operator for large clouds (n > 16384): the cloud is split over ceil(n/16384) cooperating workgroups
 * that keep their points in registers and exchange one record per iteration through `workspace`
 * (pdm_furthest_point_sampling_ws_bytes(b, n) bytes, 8-byte aligned, contents irrelevant).  Identical idx/temp.
 * The workgroups of a cloud wait for each other, so a launch holds at most pdm_fps_max_coresident_workgroups()
 * of them (compute units x resident workgroups per unit of the CURRENT device, from the runtime); the call cuts the
 * batch accordingly and takes the single-workgroup kernel where not even one cloud fits.  Every wait is bounded: a
 * workgroup that gives up raises a status word in the workspace, read back (with a stream synchronisation) by
 * pdm_furthest_point_sampling_status: *flag = 1 means the indices of that call must not be used. */
size_t pdm_furthest_point_sampling_ws_bytes(int b, int n);
int pdm_fps_max_coresident_workgroups(void);
int pdm_furthest_point_sampling_ws(void *stream, int b, int n, int m, const float *points, float *temp,
                                   int *idx, void *workspace, size_t workspace_bytes);
int pdm_furthest_point_sampling_status(void *stream, int b, int n, const void *workspace, int *flag);

/* The same operator as resumable segments (no reference counterpart; same indices): job q computes samples
 * [j0[q], j1[q]) of its own batch of b clouds, continuing from the state an earlier segment left in temp[q] (running
 * min-distances, 1e10 everywhere before the first segment) and idx[q] (samples [0, j0)).  The 1..4 jobs of a call
 * belong to different batches and run side by side in one launch; host arrays of device pointers.  1024 < n <= 16384,
 * or 16384 < n <= 131072 with workspace[q] = pdm_furthest_point_sampling_ws_bytes(b, n) bytes per job (8-byte aligned;
 * the cooperating workgroups must all be resident: njobs * b * ceil(n/16384) <= pdm_fps_max_coresident_workgroups();
 * status per job workspace as above); workspace may be null otherwise. */
int pdm_furthest_point_sampling_jobs(void *stream, int njobs, int b, int n, int m, const float *const *points,
                                     float *const *temp, int *const *idx, const int *j0, const int *j1,
                                     void *const *workspace, size_t workspace_bytes);

/* replaces three_nn_wrapper_fast              interpolate.cpp:18-26 -> interpolate_gpu.cu:16-81
 * unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) squared distances, idx (B,n,3). */
int pdm_three_nn(void *stream, int b, int n, int m, const float *unknown, const float *known,
                 float *dist2, int *idx);

/* replaces three_interpolate_wrapper_fast     interpolate.cpp:29-41 -> interpolate_gpu.cu:84-124
 * points (B,C,M), idx/weight (B,N,3) -> out (B,C,N). */
int pdm_three_interpolate(void *stream, int b, int c, int m, int n, const float *points,
                          const int *idx, const float *weight, float *out);

/* replaces three_interpolate_grad_wrapper_fast interpolate.cpp:44-56 -> interpolate_gpu.cu:127-168
 * grad_out (B,C,N) -> grad_points (B,C,M), caller-zeroed, accumulated. */
int pdm_three_interpolate_grad(void *stream, int b, int c, int n, int m, const float *grad_out,
                               const int *idx, const float *weight, float *grad_points);

/* The same backward with a caller-provided workspace (pdm_three_interpolate_grad_ws_bytes(b, n, m) bytes): the scatter is
 * first inverted into per-cloud CSR lists (known point <- its contributions), the accumulation then needs no atomics.
 * Same sums in a different fp32 order; m <= 16384 and n <= 32768, otherwise it forwards to the plain entry point. */
size_t pdm_three_interpolate_grad_ws_bytes(int b, int n, int m);
int pdm_three_interpolate_grad_ws(void *stream, int b, int c, int n, int m, const float *grad_out, const int *idx,
                                  const float *weight, float *grad_points, void *workspace, size_t workspace_bytes);

/* ---- fused forms of the same path (additions; same arithmetic, fewer passes over HBM) ------ */

/* QueryAndGroup.forward as one call (pointnet2_utils.py:241-264, use_xyz=True):
 * idx = ball_query (written, caller need not zero it), out[:,0:3] = xyz[idx] - new_xyz,
 * out[:,3:3+C] = features[:, idx].  features (B,C,N) may be NULL when c == 0.
 * out is (B, 3+C, M, nsample). */
int pdm_query_and_group(void *stream, int b, int n, int m, int c, float radius, int nsample,
                        const float *xyz, const float *new_xyz, const float *features, int *idx,
                        float *out);

/* Grid-accelerated form of pdm_ball_query: identical idx, bit for bit (same caller-zeroed contract),
 * O(candidates near each centre) instead of O(N) per centre.  Needs scratch memory, which this stateless
 * ABI takes from the caller: `workspace` of at least pdm_ball_query_grid_workspace_bytes(b, n) bytes,
 * contents irrelevant before and after the call. */
size_t pdm_ball_query_grid_workspace_bytes(int b, int n);
int pdm_ball_query_grid(void *stream, int b, int n, int m, float radius, int nsample,
                        const float *new_xyz, const float *xyz, int *idx, void *workspace,
                        size_t workspace_bytes);

/* The grid half and the query half of the call above as separate entry points, so ONE grid serves every query over the
 * same point set: both radii of an SA level (pointnet2_modules.py:37 loops its scales over the same xyz) and the
 * three_nn of the FP module whose known set it is.  Any grid gives exact results; radius_hint only sizes the cells
 * (0 = about two points per occupied cell, the nearest-neighbour setting).  workspace = pdm_ball_query_grid_workspace_bytes(b, n). */
int pdm_grid_build(void *stream, int b, int n, float radius_hint, const float *xyz, void *workspace, size_t workspace_bytes);
int pdm_ball_query_grid_prebuilt(void *stream, int b, int n, int m, float radius, int nsample, const float *new_xyz, int *idx,
                                 const void *workspace, size_t workspace_bytes);
int pdm_three_nn_grid_prebuilt(void *stream, int b, int n, int m, const float *unknown, float *dist2, int *idx,
                               const void *workspace, size_t workspace_bytes);

/* Grid-accelerated form of pdm_three_nn: identical dist2 / idx, bit for bit; needs m >= 1 and a
 * caller-provided workspace of pdm_three_nn_grid_workspace_bytes(b, m) bytes. */
size_t pdm_three_nn_grid_workspace_bytes(int b, int m);
int pdm_three_nn_grid(void *stream, int b, int n, int m, const float *unknown, const float *known,
                      float *dist2, int *idx, void *workspace, size_t workspace_bytes);

/* The reference's python glue between three_nn and three_interpolate in one launch: dist = sqrt(dist2)
 * (pointnet2_utils.py:98; optional, NULL to skip), weight = (1/(dist+1e-8)) / sum_k (1/(dist_k+1e-8))
 * (pointnet2_modules.py:154-156), same correctly-rounded fp32 operations in the same order.  rows = B*n. */
int pdm_three_nn_weights(void *stream, long long rows, const float *dist2, float *dist, float *weight);

/* The gather half of the above for a given idx (B,M,nsample): grouped xyz minus centre, grouped
 * features, concatenated on the channel axis -> out (B, 3+C, M, nsample)
 * (pointnet2_utils.py:250-257: two grouping_operation calls, the in-place subtract and torch.cat). */
int pdm_group_concat(void *stream, int b, int n, int m, int c, int nsample, const float *xyz,
                     const float *new_xyz, const float *features, const int *idx, float *out);

/* Fused inference forms of one SA scale / one FP module on fp32 MFMA (BatchNorm folded into the
 * weights by the host; pdm_ssd_amd/fused.py builds `dims`, `wpack`, `bias`).  Features are POINT-MAJOR
 * here: feat_pm (B,N,cin), out_pm (B,M,out_stride) with this scale's channels at [out_coff, out_coff+cout).
 * dims = nlayers+1 host ints: padded (multiple-of-16) widths K0, C1..CL.  wpack = per layer, for each
 * 16x16 block (mb, kb): 64 lanes x float4 = W'[16mb + (lane&15)][16kb + 4(lane>>4) + 0..3]; bias = the
 * padded per-layer shifts.  SA input channel order is [features(cin), dx, dy, dz, zero pad].
 *
 * pdm_sa_mlp_fused == pointnet2_utils.py:250-257 (group xyz/features, subtract centre, cat) +
 *                     pointnet2_modules.py:40-52 (MLP, max_pool2d over nsample) for a given idx. */
int pdm_sa_mlp_fused(void *stream, int b, int n, int m, int cin, int nsample, const float *xyz,
                     const float *new_xyz, const float *feat_pm, const int *idx, int nlayers,
                     const int *dims, const float *wpack, const float *bias, float *out_pm,
                     int out_stride, int out_coff, int cout);

/* pdm_fp_mlp_fused == pointnet2_modules.py:158-170 (three_interpolate, cat with the skip features, MLP)
 * for given idx/weight (B,n,3): known_pm (B,m,c_known), skip_pm (B,n,c_skip) or NULL -> out_pm (B,n,out_stride). */
int pdm_fp_mlp_fused(void *stream, int b, int n, int m, int c_known, int c_skip,
                     const float *known_pm, const float *skip_pm, const int *idx, const float *weight,
                     int nlayers, const int *dims, const float *wpack, const float *bias,
                     float *out_pm, int out_stride, int cout);

/* Per-row MLP on fp32 MFMA: in_pm (rows, cin) -> out_pm (rows, out_stride), same dims/wpack/bias packing.
 * relu_last = 0 leaves the last layer linear.  Serves 1x1 convolutions over point-major rows and the
 * pre-projections below. */
int pdm_rows_mlp_fused(void *stream, int rows, int cin, const float *in_pm, int nlayers, const int *dims,
                       const float *wpack, const float *bias, int relu_last, float *out_pm, int out_stride,
                       int cout);

/* Two per-row MLPs of EQUAL widths (one dims table) over the SAME rows: out_a = MLP_a(in), out_b = MLP_b(in) — the point
 * head's class and box stacks, one module with two chains on one input
 * (/root/reference/pcdet/models/dense_heads/point_head_box.py:7-60, forward :85-86).  One launch where an instantiation
 * exists (128 -> 256 -> 256 -> <= 16 over >= 8192 rows: the rows are read once), otherwise two pdm_rows_mlp_fused calls;
 * bit-identical outputs either way. */
int pdm_rows_mlp_fused_pair(void *stream, int rows, int cin, const float *in_pm, int nlayers, const int *dims,
                            const float *wpack_a, const float *bias_a, const float *wpack_b, const float *bias_b,
                            int relu_last, float *out_a, int out_stride_a, int cout_a, float *out_b, int out_stride_b,
                            int cout_b);

/* The backbone's LAST feature-propagation module and the point head's two stacks in one launch
 * (/root/reference/pcdet/models/backbones_3d/pointnet2_backbone.py:96-111 writes point_features, which
 * /root/reference/pcdet/models/dense_heads/point_head_box.py:71-76 reads straight back).  FP arguments as
 * pdm_fp_mlp_fused_pre (z = the first layer's known-feature part applied to the m known points), head arguments as
 * pdm_rows_mlp_fused_pair; out_pm receives the module's rows, out_a / out_b the stacks' outputs on them.  Bit-identical to the
 * two calls (FP through the register-resident chain kernel).  Shapes: <= 4 skip channels, dims {16, 128, 128}, hdims
 * {128, 256, 256, 16}, >= 32768 rows; PDM_E_BADARG otherwise (issue the two calls instead). */
int pdm_fp_head_fused(void *stream, int b, int n, int m, int c_skip, const float *z_pm, int z_stride, const float *skip_pm,
                      const int *idx, const float *weight, const int *dims, const float *wpack, const float *bias,
                      float *out_pm, int out_stride, int cout, const int *hdims, const float *hw_a, const float *hb_a,
                      const float *hw_b, const float *hb_b, int relu_last, float *out_a, int out_stride_a, int cout_a,
                      float *out_b, int out_stride_b, int cout_b);
int pdm_tune_fused_pair(int on);   /* 0: always two launches (A/B and tests); returns the previous setting */

/* Point head in training: target assignment + sigmoid focal loss + weighted smooth-L1 loss + d L / d predictions, per point
 * (/root/reference/pcdet/models/dense_heads/point_head_template.py:51-206 with set_ignore_flag, PointResidualCoder with mean
 * sizes box_coder_utils.py:156-179, loss_utils.py:10-141).  n_total = B * n_per_sample points in sample order; box_idx / ext_idx =
 * pdm_points_in_boxes on the boxes / the enlarged boxes; gt_boxes (B, boxes_per_sample, 8); pred_bf16 = dtype of cls_preds
 * (n_total, num_class) and box_preds (n_total, 8) (row strides in elements) AND of the gradients dcls / dbox (contiguous);
 * code_weights: 8 floats on the HOST.  labels (n_total) int64; out (3): [L_cls, L_box, #positives], L_box = NaN when a positive's
 * class exceeds the mean-size table (the reference asserts).  Three launches, no atomics on floats: bit-reproducible. */
int pdm_point_head_loss(void *stream, long long n_total, int n_per_sample, int boxes_per_sample, int num_class, int n_mean,
                        int pred_bf16, const void *cls_preds, long long cls_stride, const void *box_preds, long long box_stride,
                        const float *xyz, long long xyz_stride, const int *box_idx, const int *ext_idx, const float *gt_boxes,
                        const float *mean_size, const float *code_weights, float beta, float alpha, float gamma, float cls_weight,
                        float box_weight, long long *labels, void *dcls, void *dbox, float *out, void *workspace,
                        size_t workspace_bytes);
size_t pdm_point_head_loss_workspace_bytes(long long n_total);

/* Heat-map head in training (the dense half of the hybrid head; /root/reference/pcdet/models/dense_heads/center_head.py:100-160
 * targets, :232 clamped sigmoid, /root/reference/pcdet/utils/loss_utils.py:266-304 penalty-reduced focal loss).
 * pdm_heatmap_targets: heatmap (B, C, H, W) fp32, zeroed here, then per gt box (B, M, 8) [x y z dx dy dz heading class >= 1] a
 *   gaussian of radius max(int(gaussian_radius(dx, dy in cells, min_overlap)), min_radius), window clipped at max_radius,
 *   max-merged (/root/reference/pcdet/models/model_utils/centernet_utils.py:9-70).  cell = (x - x0) / vx / stride.
 * pdm_heatmap_focal_loss: logits (B, C, H, W) fp32 / bf16 with element strides; out[0] = - weight * S / max(peaks, 1),
 *   out[1] = - weight / max(peaks, 1), out[2] = peaks; dlogits (B, C, H, W) contiguous fp32 = d S / d logit (multiply by
 *   out[1] and the incoming gradient).  Partial sums folded in double in a fixed order: bit-reproducible. */
int pdm_heatmap_targets(void *stream, int B, int M, int C, int H, int W, const float *gt_boxes, float x0, float y0, float vx,
                        float vy, float stride, double min_overlap, int min_radius, int max_radius, float *heatmap);
size_t pdm_heatmap_focal_loss_workspace_bytes(long long n);
int pdm_heatmap_focal_loss(void *stream, int B, int C, int H, int W, const void *logits, int logits_bf16, long long sb, long long sc,
                           long long sh, long long sw, const float *heatmap, float weight, float *dlogits, float *out,
                           void *workspace, size_t workspace_bytes);

/* OPT-IN: the same three-layer per-row MLP with fp32 EMULATED on the bf16 matrix pipe — every fp32 operand split into three
 * bf16 pieces (8 + 8 + 8 significand bits), a product formed from the six leading partial products on
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation (3/8 of the fp32-MFMA pipe time; dropped terms <= 2^-24 |a b|).
 * Only dims = {128, 256, 256, 16} (the point head's stacks, point_head_box.py:7-60); `wstream` = the pre-split weights in
 * consumption order (pdm_ssd_amd/fused.py::PackedMLPx3, pdm_rows_mlp_x3_stream_bytes bytes), bias fp32 padded. */
int pdm_rows_mlp_x3(void *stream, int rows, int cin, const float *in_pm, int nlayers, const int *dims, const void *wstream,
                    size_t wstream_bytes, const float *bias, int relu_last, float *out_pm, int out_stride, int cout);
size_t pdm_rows_mlp_x3_stream_bytes(int nlayers, const int *dims);
int pdm_tune_rows_x3_wg_per_cu(int n);

/* The same SA scale / FP module with the wide part of the FIRST layer hoisted out of the per-pair (per-fine-
 * point) loop — algebraically identical, fp32 rounding order differs (tests: <= 1e-4 of the oracle):
 *   SA:  W1 [f_nb ; x_nb - c] = z[nb] + W1[:, xyz] (x_nb - c),   z = W1[:, features] f  over the n source points;
 *   FP:  W1 [sum_k w_k f_k ; s] = sum_k w_k z[idx_k] + W1[:, skip] s,   z = W1[:, known] f  over the m known points.
 * z_pm rows hold the layer's padded C1 floats at [z_coff, z_coff + C1pad) of z_stride (pdm_rows_mlp_fused with
 * relu_last = 0 and zero bias); dims[0] is the padded width of what stays in the kernel (16 for xyz; the
 * padded skip width, or 16 with zero weights when c_skip == 0). */
int pdm_sa_mlp_fused_pre(void *stream, int b, int n, int m, int nsample, const float *xyz,
                         const float *new_xyz, const float *z_pm, int z_stride, int z_coff, const int *idx,
                         int nlayers, const int *dims, const float *wpack, const float *bias, float *out_pm,
                         int out_stride, int out_coff, int cout);
int pdm_fp_mlp_fused_pre(void *stream, int b, int n, int m, int c_skip, const float *z_pm, int z_stride,
                         const float *skip_pm, const int *idx, const float *weight, int nlayers,
                         const int *dims, const float *wpack, const float *bias, float *out_pm,
                         int out_stride, int cout);

/* Neighbour-list compaction for the fused SA kernels (no reference counterpart: ball_query pads short
 * neighbourhoods with copies of the first hit, pointnet2/src/ball_query_gpu.cu:38-46, and the reference runs the
 * shared MLP over those copies; max-pool over a multiset = max-pool over the set, so they can be dropped).
 *   pdm_sa_pack: idx (B,M,nsample) int32, nsample 16 or 32 -> pack (pdm_sa_pack_rows(b,m,nsample) x 2 int32:
 *     {source row b*n + neighbour, centre b*m + j | -1 dead}) and meta (8 int32: first row of the segment classes
 *     L = 1,2,4,8,16,32, total rows, live rows).  A centre with cnt significant slots (1 + the last slot that differs
 *     from slot 0) owns 2^ceil(log2 cnt) consecutive rows; classes are tile-aligned, order = centre order.
 *   pdm_sa_mlp_packed: pdm_sa_mlp_fused (z_pm null) / pdm_sa_mlp_fused_pre (z_pm set) over that list; results are
 *     bit-identical to the unpacked entry points. */
size_t pdm_sa_pack_workspace_bytes(int b, int m);
size_t pdm_sa_pack_rows(int b, int m, int nsample);
int pdm_sa_pack(void *stream, int b, int n, int m, int nsample, const int *idx, void *workspace,
                size_t workspace_bytes, int *pack, int *meta);
/* both scales of an MSG level in one count -> scan -> fill sequence (per-scale nsample / idx / workspace / pack / meta as HOST arrays
 * of two; each workspace of workspace_bytes >= pdm_sa_pack_workspace_bytes(b, m)) */
int pdm_sa_pack_pair(void *stream, int b, int n, int m, const int *nsample, const int *const *idx, void *const *workspace,
                     size_t workspace_bytes, int *const *pack, int *const *meta);
int pdm_sa_mlp_packed(void *stream, int b, int n, int m, int cin, int nsample, const float *xyz,
                      const float *new_xyz, const float *feat_pm, const float *z_pm, int z_stride, int z_coff,
                      const int *pack, const int *meta, int nlayers, const int *dims, const float *wpack,
                      const float *bias, float *out_pm, int out_stride, int out_coff, int cout);

/* The two scales of an MSG level (pointnet2_modules.py:58-99: the same centres, two radii / MLPs) in ONE launch: per-scale
 * arguments as HOST arrays of two (nsample, z_coff, pack, meta, nlayers, dims, wpack, bias, out_coff, cout), the level's xyz /
 * new_xyz / feat_pm / z_pm / out_pm shared.  One launch where both scales map to the same kernel instantiation (the deep levels
 * of PointNet2MSG: a few hundred row tiles per scale, which two launches run one after the other on a mostly idle chip),
 * otherwise two pdm_sa_mlp_packed launches; bit-identical either way. */
int pdm_sa_mlp_packed_pair(void *stream, int b, int n, int m, int cin, const int *nsample, const float *xyz,
                           const float *new_xyz, const float *feat_pm, const float *z_pm, int z_stride, const int *z_coff,
                           const int *const *pack, const int *const *meta, const int *nlayers, const int *const *dims,
                           const float *const *wpack, const float *const *bias, float *out_pm, int out_stride,
                           const int *out_coff, const int *cout);
int pdm_tune_sa_pair(int on);          /* 0: always two launches */

/* ---- pointnet2_stack: ragged ("stacked") batches -----------------------
 * One entry per function of the reference's pointnet2_stack_cuda extension that the PointNet++ modules use
 * (pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp): points of all samples concatenated, per-sample
 * counts in int32 DEVICE arrays (*_batch_cnt, B entries, B <= 1024).  Same arithmetic as the batch entries.
 *   ball_query_wrapper_stack           ball_query.cpp      idx (M,nsample) LOCAL to the sample, caller-zeroed;
 *                                                           an empty ball stores idx[0] = -1 (ball_query_gpu.cu:66)
 *   group_points[_grad]_wrapper_stack  group_points.cpp    out (M,C,nsample); grad_features (N,C) caller-zeroed
 *   three_nn_wrapper_stack             interpolate.cpp     dist2 (N,3), idx (N,3) GLOBAL; +inf / start when the
 *                                                           sample has fewer than three known points
 *   three_interpolate[_grad]_wrapper_stack                 features (M,C) -> out (N,C); grad caller-zeroed
 *   stack_farthest_point_sampling_wrapper  sampling.cpp    GLOBAL indices, packed per sample; temp = 1e10 */
int pdm_stack_ball_query(void *stream, int B, int M, float radius, int nsample, const float *new_xyz,
                         const int *new_xyz_batch_cnt, const float *xyz, const int *xyz_batch_cnt, int *idx);
int pdm_stack_group_points(void *stream, int B, int M, int C, int nsample, const float *features,
                           const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt, float *out);
int pdm_stack_group_points_grad(void *stream, int B, int M, int C, int N, int nsample, const float *grad_out,
                                const int *idx, const int *idx_batch_cnt, const int *features_batch_cnt,
                                float *grad_features);
int pdm_stack_three_nn(void *stream, int B, int N, const float *unknown, const int *unknown_batch_cnt,
                       const float *known, const int *known_batch_cnt, float *dist2, int *idx);
int pdm_stack_three_interpolate(void *stream, int N, int C, const float *features, const int *idx,
                                const float *weight, float *out);
int pdm_stack_three_interpolate_grad(void *stream, int N, int C, const float *grad_out, const int *idx,
                                     const float *weight, float *grad_features);
/* max_n = largest per-sample point count (host value; selects the kernel instantiation) */
int pdm_stack_furthest_point_sampling(void *stream, int B, int max_n, const float *xyz, float *temp,
                                      const int *xyz_batch_cnt, int *idxs, const int *num_sampled_points);

/* ---- pointnet2_stack: voxel query and the vector-pool family (second half) ------------
 * pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:14 and :25-30 bind
 *   voxel_query_wrapper                                  voxel_query.cpp:20 / voxel_query_gpu.cu:11-91
 *   query_stacked_local_neighbor_idxs_wrapper_stack      vector_pool.cpp:63  / vector_pool_gpu.cu:125-205
 *   query_three_nn_by_stacked_local_idxs_wrapper_stack   vector_pool.cpp:19  / vector_pool_gpu.cu:19-87
 *   vector_pool_wrapper, vector_pool_grad_wrapper        vector_pool.cpp:113, :170 / vector_pool_gpu.cu:245-443
 * The reference gives out slots of the stacked outputs with atomicAdd on a cursor and has python re-run the kernel
 * with a larger buffer on overrun; here the cursor is an exclusive prefix sum of per-centre counts (centre order —
 * one of the orders the race allows, and the same on every run), exposed as a count pass and a fill pass so a
 * caller can size the buffers exactly.  pdm_stack_query_local_neighbor_idxs is the reference's one-call form
 * (count + fill into a stack of avg_length * M slots, writes stop at the capacity, *cumsum += total).
 *
 * voxel_query: idx (M, nsample) GLOBAL indices, caller-zeroed; window cells visited z outer / x inner; kept when
 *   d2 <= radius^2; the first hit fills the row; idx[0] = -1 when nothing was found.
 * local neighbours: first by index, at most nsample when nsample > 0, never more than 1000; neighbor_type 1 = ball
 *   (d2 <= r^2), otherwise cube (|l| <= r per axis); start_len (M,2) = [offset, length]; indices GLOBAL.
 * three_nn_by_local_idxs: dist2 / idxs (M, num_total_grids, 3); idx -1 and +inf for an empty list; the best is
 *   repeated into unfilled second / third slots.  stack_len = valid length of stack_neighbor_idxs.
 * vector_pool: new_features (M, num_c_out) and new_local_xyz (M, 3G) are SUMS (overwritten, no zero-fill needed;
 *   python divides by point_cnt_of_grid); input channel i folds onto i % (num_c_out / G); pooling_type 0 = sum,
 *   1 = first point of each cell; grouped_idxs (num_max_sum_points, 3) = [support idx, centre, cell] at
 *   entry_start[centre] + rank; entries past num_max_sum_points are dropped.  (num_c_out + 4G) * 4 B <= 48 KB.
 * vector_pool_grad: grad_support_features (N, C_in) caller-zeroed, float atomics (order undefined, as upstream). */
int pdm_stack_voxel_query(void *stream, int M, int Z, int Y, int X, int nsample, float radius, int z_range, int y_range,
                          int x_range, const float *new_xyz, const float *xyz, const int *new_coords,
                          const int *point_indices, int *idx);
int pdm_stack_local_neighbor_count(void *stream, const float *support_xyz, const int *xyz_batch_cnt, const float *new_xyz,
                                   const int *new_xyz_batch_cnt, int *start_len, int *cumsum, float max_neighbour_distance,
                                   int batch_size, int M, int nsample, int neighbor_type);
int pdm_stack_local_neighbor_fill(void *stream, const float *support_xyz, const int *xyz_batch_cnt, const float *new_xyz,
                                  const int *new_xyz_batch_cnt, int *stack_neighbor_idxs, const int *start_len,
                                  long long stack_capacity, float max_neighbour_distance, int batch_size, int M, int nsample,
                                  int neighbor_type);
int pdm_stack_query_local_neighbor_idxs(void *stream, const float *support_xyz, const int *xyz_batch_cnt,
                                        const float *new_xyz, const int *new_xyz_batch_cnt, int *stack_neighbor_idxs,
                                        int *start_len, int *cumsum, int avg_length_of_neighbor_idxs,
                                        float max_neighbour_distance, int batch_size, int M, int nsample, int neighbor_type);
int pdm_stack_three_nn_by_local_idxs(void *stream, const float *support_xyz, const float *new_xyz_grid_centers,
                                     int *new_xyz_grid_idxs, float *new_xyz_grid_dist2, const int *stack_neighbor_idxs,
                                     const int *start_len, long long stack_len, int M, int num_total_grids);
/* entry_cnt (M) = entries each centre records, entry_start (M) = their exclusive prefix, *total += the sum */
int pdm_stack_vector_pool_count(void *stream, const float *support_xyz, const int *xyz_batch_cnt, const float *new_xyz,
                                const int *new_xyz_batch_cnt, int *entry_start, int *entry_cnt, int *total, int num_grid_x,
                                int num_grid_y, int num_grid_z, float max_neighbour_distance, int batch_size, int M,
                                int nsample, int neighbor_type, int pooling_type);
int pdm_stack_vector_pool(void *stream, const float *support_xyz, const float *support_features, const int *xyz_batch_cnt,
                          const float *new_xyz, float *new_features, float *new_local_xyz, const int *new_xyz_batch_cnt,
                          int *point_cnt_of_grid, int *grouped_idxs, const int *entry_start, int num_grid_x, int num_grid_y,
                          int num_grid_z, float max_neighbour_distance, int batch_size, int M, int num_c_in, int num_c_out,
                          int use_xyz, int num_max_sum_points, int nsample, int neighbor_type, int pooling_type);
int pdm_stack_vector_pool_grad(void *stream, const float *grad_new_features, const int *point_cnt_of_grid,
                               const int *grouped_idxs, float *grad_support_features, int N, int M, int num_c_out,
                               int num_c_in, int num_total_grids, int num_entries);

/* ---- training-mode BatchNorm + ReLU as one operator (configs[3], the shared MLPs' Conv/Linear -> BN -> ReLU triples:
 * pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:19-55, models/dense_heads/point_head_template.py:35-48) ----
 * dtype 0 = fp32, 1 = bf16 activations (statistics, gamma/beta and gradients of gamma/beta always fp32).
 * layout 0: x is (n rows, C) with the channel fastest, C a multiple of 4 (fp32) / 8 (bf16); L ignored.
 * layout 1: x is (n, C, L) with the position fastest, L a multiple of 4 / 8.
 * forward: batch mean / biased variance per channel, y = [relu]((x - mean) * invstd * gamma + beta); running_mean /
 *   running_var (may be null) updated as torch.nn.BatchNorm does (momentum, unbiased variance); coef (4, C) fp32 =
 *   [mean | invstd | gamma*invstd | beta] is what the backward needs besides x.
 * backward: dx (same type and layout as x), grads (4, C) fp32 = [dgamma | dbeta | p | q] (dx = gamma invstd (g - p - (x - mean) q)).
 * partial: workspace of pdm_bn_parts(layout, n, C, L) * C * 2 floats (slice sums, folded in double: reproducible). */
int pdm_bn_parts(int layout, long long n, int C, long long L);
int pdm_bn_relu_forward(void *stream, int dtype, int layout, long long n, int C, long long L, const void *x, void *y,
                        const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                        float *running_var, float *coef, float *partial, int relu);
/* the same forward with the statistics already taken by the producer of x (pdm_tg_gemm_nt's `stats`: [parts][C][2] column sums
 * of x and x^2): finalize + apply only, rows x C layout */
int pdm_bn_relu_forward_stats(void *stream, int dtype, long long n, int C, const void *x, void *y, const float *gamma,
                              const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                              float *coef, const float *partial, int parts, int relu);
/* statistics only (rows x C; dtype 0 / 1): reduce + finalize -> coef (4, C) and the running statistics, no normalised tensor (a
 * consumer applies BatchNorm + ReLU while it reads x); partial: pdm_bn_parts(0, n, C, 1) * C * 2 floats */
int pdm_bn_forward_coef(void *stream, int dtype, long long n, int C, const void *x, const float *gamma, const float *beta, float eps,
                        float momentum, float *running_mean, float *running_var, float *coef, float *partial);
/* finalize only: coef (4, C) from the producer's sums + running statistics (the consumer normalises while it reads x) */
int pdm_bn_finalize_stats(void *stream, long long n, int C, const float *gamma, const float *beta, float eps, float momentum,
                          float *running_mean, float *running_var, float *coef, const float *partial, int parts);
int pdm_bn_relu_backward(void *stream, int dtype, int layout, long long n, int C, long long L, const void *x, const void *dy,
                         void *dx, const float *coef, float *grads, float *partial, int relu);
/* its two halves: `_stats` leaves grads (4, C) = [dgamma | dbeta | p | q] and writes no dx (a consumer forms dx while it reads
 * dy and x: pdm_tg_gemm_nt_dy); `_apply` writes dx = scale (dy [bn(x) > 0] - p - (x - mean) q) from grads already there */
int pdm_bn_relu_backward_stats(void *stream, int dtype, int layout, long long n, int C, long long L, const void *x, const void *dy,
                               const float *coef, float *grads, float *partial, int relu);
int pdm_bn_relu_backward_apply(void *stream, int dtype, int layout, long long n, int C, long long L, const void *x, const void *dy,
                               void *dx, const float *coef, float *grads, int relu);
/* finalize only: grads (4, C) from [parts][C][2] sums (sum g, sum g xhat) a producer of dy has already taken in its epilogue
 * (pdm_tg_gemm_nt_bs / pdm_tg_gemm_nt_dy_bs): pdm_bn_relu_backward_stats without its pass over dy and x */
int pdm_bn_finalize_bwd_stats(void *stream, long long n, int C, const float *coef, float *grads, const float *partial, int parts);

/* The tail of an SA scale in training — BatchNorm + ReLU + max over the ns neighbours of each group
 * (pointnet2_modules.py:46-52: the last (BatchNorm2d, ReLU) of the shared MLP, then F.max_pool2d over nsample) — as ONE
 * operator.  x is (G groups, ns <= 255, C) with the channel fastest; y (G, C).  relu(bn(.)) is monotone in x, so one pass
 * over x gives the statistics and each group's max / min (+ first index), the pooled output is the function of that
 * extreme, and the normalised tensor is never written; backward = sums over the pooled tensors + one pass for dx.
 * xmax / xmin (G, C, x's type) and imax / imin (G, C, bytes) are the forward's record for the backward.
 * pdm_bn_pool_parts: slices of `partial` ((parts, C, 2) floats). */
int pdm_bn_pool_parts(int dtype, long long G, int C);
int pdm_bn_relu_pool_forward(void *stream, int dtype, long long G, int ns, int C, const void *x, void *y, void *xmax, void *xmin,
                             unsigned char *imax, unsigned char *imin, const float *gamma, const float *beta, float eps,
                             float momentum, float *running_mean, float *running_var, float *coef, float *partial, int relu);
/* the forward when the producer of x (pdm_tg_gemm_nt_pool) has already left the column sums ([parts][C][2]) and every group's
 * max / min: finalize + pooled output only; bf16 (dtype 1) */
int pdm_bn_relu_pool_forward_kept(void *stream, int dtype, long long G, int ns, int C, void *y, const void *xmax, const void *xmin,
                                  const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                  float *running_var, float *coef, const float *partial, int parts, int relu);
int pdm_bn_relu_pool_backward(void *stream, int dtype, long long G, int ns, int C, const void *x, const void *dy, void *dx,
                              const void *xmax, const void *xmin, const unsigned char *imax, const unsigned char *imin,
                              const float *coef, float *grads, float *partial, int relu);

/* ---- hybrid head (north_star configs[2]: "backbone + PDM neck + hybrid head"; no reference source: SURVEY.md F1) ----
 * Depthwise 3x3 convolution + folded BatchNorm + ReLU over the neck's channels-last BEV grid, the context stage of
 * the heat-map head (pdm_ssd_amd/dense_heads/pdm_heatmap_head.py); the point head's MLPs and the heat-map head's 1x1
 * stack run through pdm_rows_mlp_fused.  in / out (B, H, W, C) fp32, w (9, C) tap-major, shift (C); C % 4 == 0. */
int pdm_bev_depthwise3x3(void *stream, int B, int H, int W, int C, const float *in, const float *w, const float *shift,
                         float *out, int relu);
/* The heat-map head's inference stack in ONE launch: out[cell] = MLP(relu(depthwise3x3(map)[cell] + shift)), the
 * depthwise convolution formed on the fly as the prologue of the register-resident row MLP (csrc/rows_chain.hip); same
 * arithmetic order as pdm_bev_depthwise3x3 followed by pdm_rows_mlp_fused (bit-identical).  Packed weights as for
 * pdm_rows_mlp_fused.  Instantiated for C = 128 -> 64 -> 64 -> <= 16; PDM_E_BADARG for other widths. */
int pdm_bev_head_fused(void *stream, int B, int H, int W, int C, const float *map, const float *dw_w, const float *dw_shift,
                       int nlayers, const int *dims, const float *wpack, const float *bias, int relu_last, float *out_pm,
                       int out_stride, int cout);
/* Point head epilogue (point_head_box.py:97-113 in eval mode): score = sigmoid(max class logit), box = the
 * PointResidualCoder decoding (box_coder_utils.py:188-222, use_mean_size) of the code with the arg-max class's mean size,
 * for every point in one pass.  cls (N, cls_stride), code (N, code_stride >= 8, rows 16-byte aligned), pts (N, pts_stride >= 3),
 * mean_size (num_class, 3) -> boxes (N, 7), scores (N). */
int pdm_point_head_decode(void *stream, long long n, int num_class, const float *cls, int cls_stride, const float *code,
                          int code_stride, const float *pts, int pts_stride, const float *mean_size, float *boxes, float *scores);
/* Weight gradient of that convolution (training): gw (9, C), zeroed by the caller, += sum over cells of gout * in[tap].
 * The data gradient is pdm_bev_depthwise3x3 on gout with the nine taps mirrored. */
int pdm_bev_depthwise3x3_wgrad(void *stream, int B, int H, int W, int C, const float *in, const float *gout, float *gw);
/* the pair above with ONE side of the map held in bf16 (training under bf16 autocast; fp32 arithmetic, one rounding to nearest
 * even): forward fp32 -> bf16 (out_bf16), data gradient bf16 -> fp32 (in_bf16), weight gradient with a bf16 output gradient */
int pdm_bev_depthwise3x3_t(void *stream, int B, int H, int W, int C, const void *in, int in_bf16, const float *w, const float *shift,
                           void *out, int out_bf16, int relu);
int pdm_bev_depthwise3x3_wgrad_t(void *stream, int B, int H, int W, int C, const float *in, const void *gout, int gout_bf16, float *gw);

/* ---- rotated-box IoU / NMS ---------------------------------------------
 * One entry per function of the reference's iou3d_nms_cuda extension (pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp):
 *   boxes_overlap_bev_gpu / boxes_iou_bev_gpu        iou3d_nms.cpp:55,96     (na, nb) areas / BEV IoUs
 *   boxes_aligned_overlap_bev_gpu, paired_boxes_overlap_bev_gpu  :37,76      element-wise overlap of two box lists
 *   nms_gpu / nms_normal_gpu                         iou3d_nms.cpp:137,186   greedy suppression of score-sorted boxes
 * Boxes are rows of 7 floats [x, y, z, dx, dy, dz, heading].  pdm_nms differs from the reference in mechanism only:
 * the suppression mask is reduced on the device; keep (n) int64 and *num_out are DEVICE buffers, workspace >=
 * pdm_nms_workspace_bytes(n). */
int pdm_boxes_overlap_bev(void *stream, int na, const float *boxes_a, int nb, const float *boxes_b, float *out);
int pdm_boxes_iou_bev(void *stream, int na, const float *boxes_a, int nb, const float *boxes_b, float *out);
int pdm_boxes_aligned_overlap_bev(void *stream, int n, const float *boxes_a, const float *boxes_b, float *out);
size_t pdm_nms_workspace_bytes(int n);
int pdm_nms(void *stream, int n, const float *boxes, float thresh, int normal, void *workspace,
            size_t workspace_bytes, long long *keep, int *num_out);
/* roiaware_pool3d's points_in_boxes_gpu (roiaware_pool3d.cpp / roiaware_pool3d_kernel.cu:313-336): boxes (B,T,7),
 * pts (B,M,3) -> box_idx (B,M) = first containing box of the sample's list, -1 for background (every entry written). */
int pdm_points_in_boxes(void *stream, int B, int T, int M, const float *boxes, const float *pts, int *box_idx);

/* ---- input path -------------------------------------------------------
 * sample_points (pcdet/datasets/processor/data_processor.py:182-212) + the batch-index column of collate_batch
 * (pcdet/datasets/dataset.py:237-244) for B raw clouds resident in HBM: raw (sum counts, C) rows [x, y, z, ...],
 * counts (B) int32 on the device, out (B * num_points, 1 + C) rows [cloud, x, y, z, ...]; choice (B * num_points)
 * int32 or NULL receives the chosen raw row per output row.  The random draw is defined by counter-based hashes of
 * (seed, cloud, row) — DESIGN.md section 10 — not by numpy's RNG.  num_points <= 16384, 3 <= C <= 16. */
int pdm_sample_points(void *stream, int B, int num_points, unsigned seed, int C, const float *raw,
                      const int *counts, float *out, int *choice);

/* ---- rows form of the FP module's input for the training path (csrc/interpolate.hip) --------------------------------
 * out (B, n, ld) bf16 = [ three_interpolate(known, idx, weight) (C2) | skip (C1) | zeros ]: the reference's
 * cat([interpolated, unknow_feats], dim=1) (pointnet2_modules.py:158-165) written once as the rows the bf16 layers read, each
 * element the fp32 value (pinned fma order of pdm_three_interpolate) rounded to nearest even.  known (B, m, C2) and skip
 * (B, n, C1) are point-major rows, fp32 or bf16.  Backward: dx (B, n, ld) bf16 -> dknown (B, m, C2) fp32 through an inverted
 * (CSR) index, no atomics; the skip gradient is dx[..., C2 : C2 + C1].  workspace: pdm_three_interpolate_grad_ws_bytes(b, n, m). */
int pdm_interp_concat_rows(void *stream, int b, int n, int m, int c2, int c1, int ld, const void *known, int known_bf16,
                           const void *skip, int skip_bf16, const int *idx, const float *weight, void *out);
int pdm_interp_concat_rows_grad(void *stream, int b, int n, int m, int c2, int ld, const void *dx, const int *idx,
                                const float *weight, float *dknown, void *workspace, size_t workspace_bytes);
/* the same, the gradient of the known rows written as fp32 (out_bf16 = 0) or bf16 (1: the rounding a cast of the fp32 result would do) */
int pdm_interp_concat_rows_grad_out(void *stream, int b, int n, int m, int c2, int ld, const void *dx, const int *idx,
                                    const float *weight, void *dknown, int out_bf16, void *workspace, size_t workspace_bytes);

/* ---- bf16 contractions of the training path (csrc/train_gemm.hip) ---------------------------------------------------
 * The shared MLPs' 1x1 convolutions / Linear layers in TRAINING (reference: torch Conv2d / Linear inside
 * pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:91-97 and models/dense_heads/point_head_template.py:35-48)
 * over channels-last rows: bf16 operands, fp32 accumulation on v_mfma_f32_32x32x16_bf16, ONE rounding of the result.
 * All strides in elements and multiples of 8, K and N multiples of 8, pointers 16-byte aligned. */
/* Y (R, N) bf16 = X (R, K) . W (N, K)^T [+ bias (N) fp32, rounded to bf16 first].  stats: null, or
 * (pdm_tg_stats_parts(R, N), N, 2) fp32 = per persistent slot the column sums of y and y^2 of the ROUNDED outputs (BatchNorm
 * statistics without another pass over Y; pdm_bn_relu_forward_stats folds the parts).  The data gradient is the same call on
 * the transposed weights. */
int pdm_tg_stats_parts(long long rows, int N);
/* x_bn_coef: null, or (4, K) fp32 [mean | invstd | gamma invstd | beta] from pdm_bn_finalize_stats: X holds the PRE-BatchNorm
 * outputs of the layer before and is read through bf16(relu((x - mean) scale + beta)) — that layer's BatchNorm + ReLU without
 * a pass (or a tensor) of its own; bit for bit what pdm_bn_relu_forward would have written. */
int pdm_tg_gemm_nt(void *stream, long long R, int K, int N, const void *X, long long ldx, const void *W, long long ldw,
                   void *Y, long long ldy, const float *bias, float *stats, const float *x_bn_coef);
/* Data gradient straight behind a BatchNorm + ReLU backward (the autograd rule of torch.nn.BatchNorm1d/2d + ReLU in
 * pointnet2_modules.py:91-97's stacks, fused into the contraction that consumes it): dX (R, N) = dY (R, K) . W (N, K)^T with
 * dY = scale (dZ [bn(Yp) > 0] - p - (Yp - mean) q) formed from dZ and Yp (R, K) while the operand is staged, and written to
 * dYout (R, K) on the way for the layer's weight gradient.  coef (4, K) from pdm_bn_finalize_stats, grads (4, K) from
 * pdm_bn_relu_backward_stats.  dYout is bit for bit pdm_bn_relu_backward's dx. */
int pdm_tg_gemm_nt_dy(void *stream, long long R, int K, int N, const void *dZ, long long lddz, const void *Yp, long long ldyp,
                      const void *W, long long ldw, void *dX, long long lddx, void *dYout, long long lddy, const float *coef,
                      const float *grads);
/* The two products above when their result IS the gradient of relu(bn(Bx)) — the data gradient of the layer that follows a
 * BatchNorm + ReLU in a stack (pointnet2_modules.py:91-97, point_head_template.py:35-48): the epilogue also leaves the statistics
 * that BatchNorm's backward needs, per slot and column the sums of g = y [bn(Bx) > 0] and of g (Bx - mean) invstd over the ROUNDED
 * outputs, in bstats (parts, N, 2) fp32 with parts = pdm_tg_stats_parts(R, N) / pdm_tg_dy_stats_parts(R, N) — the reduce pass of
 * pdm_bn_relu_backward_stats (a second read of Y and Bx) disappears; pdm_bn_finalize_bwd_stats folds the parts.
 * Bx (R, N) bf16 = that BatchNorm's input, bcoef (4, N) = its coefficients from pdm_bn_finalize_stats. */
int pdm_tg_gemm_nt_bs(void *stream, long long R, int K, int N, const void *X, long long ldx, const void *W, long long ldw,
                      void *Y, long long ldy, const void *Bx, long long ldbx, const float *bcoef, float *bstats);
/* pdm_tg_gemm_nt for the last layer of an SA scale (pointnet2_modules.py:46-52: ... -> BatchNorm -> ReLU -> max over nsample): rows
 * g ns .. g ns + ns - 1 are group g; the epilogue also leaves every group's per-channel max / min of the rounded outputs and the first
 * index attaining them (xmax, xmin (R / ns, N) bf16; imax, imin (R / ns, N) bytes) — pdm_bn_relu_pool_forward's statistics pass
 * without its read of Y; pdm_bn_relu_pool_forward_kept finishes the operator.  ns a power of two, 4 .. 128, dividing R. */
int pdm_tg_gemm_nt_pool(void *stream, long long R, int K, int N, const void *X, long long ldx, const void *W, long long ldw,
                        void *Y, long long ldy, const float *bias, float *stats, const float *x_bn_coef, int ns, void *xmax,
                        void *xmin, unsigned char *imax, unsigned char *imin);
int pdm_tg_dy_stats_parts(long long rows, int N);
int pdm_tg_gemm_nt_dy_bs(void *stream, long long R, int K, int N, const void *dZ, long long lddz, const void *Yp, long long ldyp,
                         const void *W, long long ldw, void *dX, long long lddx, void *dYout, long long lddy, const float *coef,
                         const float *grads, const void *Bx, long long ldbx, const float *bcoef, float *bstats);
size_t pdm_tg_wgrad_ws_bytes(long long R, int K, int N);
/* dW (N, K) fp32 (+)= dY (R, N)^T . X (R, K): row slabs summed in a fixed order (bit-reproducible) */
int pdm_tg_wgrad(void *stream, long long R, int K, int N, const void *dY, long long ldy, const void *X, long long ldx, float *dW,
                 int accumulate, void *workspace, size_t workspace_bytes, const float *x_bn_coef);
/* out (N) fp32 = column sums of Y (R, N) bf16 — the bias gradient; N a multiple of 8, <= 512; fixed summation order */
size_t pdm_tg_colsum_ws_floats(long long R, int N);
int pdm_tg_colsum(void *stream, long long R, int N, const void *Y, long long ld, float *out, float *scratch);
/* W (N, K) fp32 -> bf16 Wb (N, ldb) and / or its transpose Wt (K, ldt); either may be null; pad columns zero */
int pdm_tg_pack_weight(void *stream, int N, int K, const float *W, void *Wb, int ldb, void *Wt, int ldt);
/* the pair a layer needs, every element written: Wb (rows_to, cols_to) = W zero padded, Wt (cols_to, rows_to) = its transpose */
int pdm_tg_pack_weight_pair(void *stream, int N, int K, const float *W, void *Wb, void *Wt, int rows_to, int cols_to);
/* the same for many layers in one launch.  jobs: njobs records of 48 bytes in DEVICE memory,
 * { const float *W; void *Wb; void *Wt; int N, K, rows_to, cols_to; long long first_block; } with first_block[0] = 0,
 * first_block[j + 1] = first_block[j] + ceil(rows_to[j] * cols_to[j] / 256); total_blocks = their sum */
int pdm_tg_pack_weight_many(void *stream, int njobs, const void *jobs, long long total_blocks);


/* ---- diagnostics (process-global tuning switches used by tools/diag/ A/B measurements; every setting gives
 * identical results; each returns the previous value; not for production callers) ------------------------- */
int pdm_tune_fps_variant(int v);        /* 8192 < n <= 16384: 0 pruned 1024x16 (default), 3 pruned 512x32, 1 / 2 unpruned */
int pdm_tune_fused_waves(int w);        /* channel-split waves per workgroup: 0 heuristic, 1/2/4/8 */
int pdm_tune_fused_tiles(int t);        /* 16-position tiles per group: 0 heuristic, 1, 2 */
int pdm_tune_fused_groups(int n);       /* position groups per workgroup: 0 heuristic, 1/2/4/8 */
int pdm_tune_fused_wg_per_cu(int n);    /* grid cap = 256 CUs x n workgroups */
int pdm_tune_fused_lds_cap(int bytes);  /* LDS a two-tile workgroup may take (<= 160 KB) */
int pdm_tune_fused_reg(int on);         /* register-resident SA form for small scales */
int pdm_tune_fused_gemm(int on);        /* LDS-tiled GEMM for single-layer rows / two-layer FP with tiny skip */
int pdm_tune_fused_chain(int on);       /* register-resident chain kernels (rows_chain.hip) for many-row MLPs and FP modules 1-2 */
int pdm_tune_fused_swz(int on);         /* XOR-swizzled LDS tiles in the general chain kernels */
int pdm_tune_bq_quad(int form);         /* grid ball query: 1 four centres per wave (default), 2 one centre per lane, 3 per cloud by density (lane form for sparse, quad form for dense clouds), 0 one wave per centre; same indices */
int pdm_tune_bq_dense_ppc(int hundredths); /* form 3: points per occupied grid cell (x 100) from which a cloud counts as dense (default 300) */
int pdm_tune_grid_split(int min_n);     /* search-grid build: clouds of >= min_n points scatter from many workgroups (default 0 = never: measured slower) */
int pdm_tune_bq_small_waves(int w);     /* exhaustive ball query with <= 32768 centres in the call: waves per workgroup, 4 or 16 (default) */
int pdm_tune_bq_cpw(int centres);       /* lane form: centres per wave, 16 (default) / 32 / 64 */
int pdm_tune_bq_heavy(int candidates);  /* lane form: centres with more candidates than this take the whole-wave path (default 96) */
int pdm_tune_group_lds_floor(int bytes); /* LDS-staged group_points: request at least this much LDS per workgroup (caps the workgroups per CU so that kernels of other streams find wave slots beside a streaming gather); 0 = what the rows need */
int pdm_tune_group_rows(int packed);    /* group_points LDS form: 0 heuristics; variant (1 rows kernel, 2 round-2 kernel, 3 rows kernel in plain unit order) | rows per workgroup << 4 | parts of L << 8 | threads / 256 << 16 | index quads per lane and pass << 20 */
int pdm_tune_rows_chain_wg_per_cu(int n); /* grid cap of the many-row chain kernels = 256 CUs x n workgroups (default 12; 2 resident) */
int pdm_tune_fp_head_tiles(int n);           /* pdm_fp_head_fused: consecutive 64-row tiles per workgroup (default 2) */
int pdm_tune_rows_chain_dw_wg_per_cu(int n); /* the same for pdm_bev_head_fused (default 2: every workgroup resident from the start) */
int pdm_tune_rows_chain_xcd(int on);   /* heat-map chain kernel: contiguous patch range per XCD (default) / launch order */
int pdm_tune_fp_chain_pad_lds(int bytes); /* diagnostic: extra LDS per workgroup of the FP chain kernel (occupancy experiments) */

/* count device-to-device copies dst[k] <- src[k] (bytes[k] each; host arrays) in one launch per 48 buffers.
 * Plumbing for the stream pipeline's hand-over buffers, not a reference operator. */
int pdm_copy_many(void *stream, int count, void *const *dst, const void *const *src, const size_t *bytes);
/* The same with device-side lengths: where dyn_count[k] != NULL only the first *dyn_count[k] * dyn_unit[k] bytes of buffer
 * k are copied (the count is read when the kernel runs): worst-case-sized buffers with a device-computed number of
 * live rows, e.g. the row lists of pdm_sa_pack (count = &meta[6], unit = 8). */
int pdm_copy_many_dyn(void *stream, int count, void *const *dst, const void *const *src, const size_t *bytes,
                      const int *const *dyn_count, const unsigned *dyn_unit);

/* ---- PDM neck (build-defined spec, DESIGN.md "PDM spec"; no reference source exists) ------- */

/* Multi-centre scatter-add of dilated, SH x Gaussian weighted point features into a BEV grid.
 * xyz (B,P,3), feat (B,P,C), sh (B,P,(degree+1)^2), inv2s2 (B,P); origin/cell/inv_cell are
 * 3 floats each passed by value; grid dims W (x), H (y), D (z); dilation kx,ky,kz odd.
 * layout 1: grid is (B,H,W,C*D) (channels-last storage of the logical (B,C*D,H,W) tensor,
 * inner index c*D+z); layout 0: (B,C*D,H,W) contiguous.  wsum (B,H,W,D).  Both caller-zeroed. */
int pdm_scatter_bev(void *stream, int B, int P, int C, int degree, const float *xyz,
                    const float *feat, const float *sh, const float *inv2s2, float ox, float oy,
                    float oz, float cx, float cy, float cz, float icx, float icy, float icz, int W,
                    int H, int D, int kx, int ky, int kz, int layout, float *grid, float *wsum);

/* grid[cell] /= wsum[cell] where |wsum| > eps */
int pdm_bev_normalize(void *stream, int B, int C, int W, int H, int D, int layout, float eps,
                      float *grid, const float *wsum);

/* Gather form of pdm_scatter_bev (+ pdm_bev_normalize when normalize != 0) for layout 1: every cell of
 * grid (B,H,W,C*D) and wsum (B,H,W,D) is WRITTEN (no zero-fill by the caller), without atomics and in a fixed
 * summation order (ascending point index), so the result is bitwise reproducible.  workspace >=
 * pdm_gather_bev_workspace_bytes(...) bytes. */
size_t pdm_gather_bev_workspace_bytes(int B, int P, int W, int H, int kx, int ky);
int pdm_gather_bev(void *stream, int B, int P, int C, int degree, const float *xyz, const float *feat,
                   const float *sh, const float *inv2s2, float ox, float oy, float oz, float cx, float cy,
                   float cz, float icx, float icy, float icz, int W, int H, int D, int kx, int ky, int kz,
                   int normalize, float eps, float *grid, float *wsum, void *workspace, size_t workspace_bytes);

/* Backward of pdm_bev_normalize for layout 1 (channels-last): y = the normalised grid, dy its gradient ->
 * dx (B,H,W,C*D) and dwsum (B,H,W,D), both fully written.  dx may be NULL (dwsum only: pdm_scatter_bev_grad_normalized
 * applies the division on its own reads of dy). */
int pdm_bev_normalize_grad(void *stream, int B, int C, int W, int H, int D, float eps, const float *y,
                           const float *wsum, const float *dy, float *dx, float *dwsum);

/* backward of pdm_scatter_bev w.r.t. feat, sh, inv2s2 (outputs fully written, no zero-fill needed);
 * dwsum may be NULL. */
int pdm_scatter_bev_grad(void *stream, int B, int P, int C, int degree, const float *xyz,
                         const float *feat, const float *sh, const float *inv2s2, float ox,
                         float oy, float oz, float cx, float cy, float cz, float icx, float icy,
                         float icz, int W, int H, int D, int kx, int ky, int kz, int layout,
                         const float *dgrid, const float *dwsum, float *dfeat, float *dsh,
                         float *dinv2s2);
/* The same for the NORMALISED map y = grid / wsum (where |wsum| > eps): dy = dL/dy as it arrives, wsum (B,H,W,D) as the
 * forward left it, dwsum = pdm_bev_normalize_grad's second output (required).  No dL/dgrid tensor is formed. */
int pdm_scatter_bev_grad_normalized(void *stream, int B, int P, int C, int degree, const float *xyz,
                                    const float *feat, const float *sh, const float *inv2s2, float ox,
                                    float oy, float oz, float cx, float cy, float cz, float icx, float icy,
                                    float icz, int W, int H, int D, int kx, int ky, int kz, int layout,
                                    const float *dy, const float *wsum, float eps, const float *dwsum,
                                    float *dfeat, float *dsh, float *dinv2s2);

/* SURVEY.md section 8(f) row N4 -- score-ranked (instance-aware) sampling instead of FPS.  The sampling code of the
 * PDM-SSD / IA-SSD lineage is absent from the reference snapshot; its analog there is torch.topk over per-point scores.
 * Build-defined total order: score descending on the order-preserving integer image of the float (-0.0 < +0.0, NaN of
 * either sign ranks above +inf as in torch.topk), equal images by lower index.
 * scores (B,n) f32 -> idx (B,k) int32, idx[b,r] = index of the r-th ranked point; k <= n, k <= 16384. */
int pdm_topk_sampling(void *stream, int b, int n, int k, const float *scores, int *idx);

/* Diagnostics: *slot = the device's constant-rate counter (100 MHz) when `stream` reaches this point. */
int pdm_mark_time(void *stream, unsigned long long *slot);

#ifdef __cplusplus
}
#endif
#endif /* PDMSSD_HIP_H */
