// Ragged ("stacked") variants of the PointNet++ operators: the points of all samples are concatenated and
// per-sample counts arrive in *_batch_cnt device arrays (reference: pcdet/ops/pointnet2/pointnet2_stack/src/*).
// Same arithmetic and index semantics as the batch kernels (pinned squared distance, strict <, first-by-index).
// SURVEY.md section 8(f) row N3.  FPS for stacked batches lives in sampling.hip (ragged mode of the same kernels).
#include "common.h"

namespace pdm {

// ---- ball query: one wave per centre, 64 candidates per step, ballot-ordered writes, early exit ----------
__global__ __launch_bounds__(256) void stack_ball_query_kernel(int B, int M, float radius, int nsample,
                                                               const float *__restrict__ new_xyz,
                                                               const int *__restrict__ new_cnt,
                                                               const float *__restrict__ xyz,
                                                               const int *__restrict__ xyz_cnt, int *__restrict__ idx) {
    __shared__ int s_new[ST_MAXB + 1], s_xyz[ST_MAXB + 1];
    build_prefix(B, new_cnt, s_new);
    if (threadIdx.x == 64) {   // a second wave builds the other table
        int acc = 0;
        for (int k = 0; k < B; ++k) { s_xyz[k] = acc; acc += xyz_cnt[k]; }
        s_xyz[B] = acc;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pt >= M) return;
    const int bs = sample_of(pt, B, s_new);
    const float *__restrict__ src = xyz + (size_t)s_xyz[bs] * 3;
    const int n = s_xyz[bs + 1] - s_xyz[bs];
    const float r2 = __fmul_rn(radius, radius);
    const float nx = new_xyz[(size_t)pt * 3], ny = new_xyz[(size_t)pt * 3 + 1], nz = new_xyz[(size_t)pt * 3 + 2];
    int *__restrict__ out = idx + (size_t)pt * nsample;
    int cnt = 0, first = -1;
    for (int base = 0; base < n && cnt < nsample; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n) hit = sqdist(nx - src[(size_t)k * 3], ny - src[(size_t)k * 3 + 1], nz - src[(size_t)k * 3 + 2]) < r2;
        const unsigned long long mask = __ballot(hit);
        if (mask == 0) continue;
        if (cnt == 0) first = base + __ffsll((long long)mask) - 1;
        const int rank = cnt + __popcll(mask & ((1ull << lane) - 1));
        if (hit && rank < nsample) out[rank] = k;
        cnt += __popcll(mask);
    }
    if (cnt == 0) {
        if (lane == 0) out[0] = -1;   // ball_query_gpu.cu:66; the python glue zeroes the row and returns the mask
    } else {
        for (int l = cnt + lane; l < nsample; l += 64) out[l] = first;   // :56-60 padding with the first hit
    }
}

// ---- grouping: out (M, C, nsample), one thread per output element (consecutive threads, consecutive stores) --
__global__ __launch_bounds__(256) void stack_group_points_kernel(int B, int M, int C, int nsample,
                                                                 const float *__restrict__ features,
                                                                 const int *__restrict__ feat_cnt,
                                                                 const int *__restrict__ idx,
                                                                 const int *__restrict__ idx_cnt, float *__restrict__ out) {
    __shared__ int s_idx[ST_MAXB + 1], s_feat[ST_MAXB + 1];
    build_prefix(B, idx_cnt, s_idx);
    if (threadIdx.x == 64) {
        int acc = 0;
        for (int k = 0; k < B; ++k) { s_feat[k] = acc; acc += feat_cnt[k]; }
        s_feat[B] = acc;
    }
    __syncthreads();
    const long long total = (long long)M * C * nsample;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(e % nsample);
        const int c = (int)((e / nsample) % C);
        const int pt = (int)(e / nsample / C);
        const int bs = sample_of(pt, B, s_idx);
        out[e] = features[((size_t)s_feat[bs] + idx[(size_t)pt * nsample + s]) * C + c];
    }
}

__global__ __launch_bounds__(256) void stack_group_points_grad_kernel(int B, int M, int C, int nsample,
                                                                      const float *__restrict__ grad_out,
                                                                      const int *__restrict__ idx,
                                                                      const int *__restrict__ idx_cnt,
                                                                      const int *__restrict__ feat_cnt,
                                                                      float *__restrict__ grad_features) {
    __shared__ int s_idx[ST_MAXB + 1], s_feat[ST_MAXB + 1];
    build_prefix(B, idx_cnt, s_idx);
    if (threadIdx.x == 64) {
        int acc = 0;
        for (int k = 0; k < B; ++k) { s_feat[k] = acc; acc += feat_cnt[k]; }
        s_feat[B] = acc;
    }
    __syncthreads();
    const long long total = (long long)M * C * nsample;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(e % nsample);
        const int c = (int)((e / nsample) % C);
        const int pt = (int)(e / nsample / C);
        const int bs = sample_of(pt, B, s_idx);
        atomicAdd(grad_features + ((size_t)s_feat[bs] + idx[(size_t)pt * nsample + s]) * C + c, grad_out[e]);
    }
}

// ---- three nearest known points of the same sample (global indices) ------------------------------------------
__global__ __launch_bounds__(256) void stack_three_nn_kernel(int B, int N, const float *__restrict__ unknown,
                                                             const int *__restrict__ unk_cnt,
                                                             const float *__restrict__ known,
                                                             const int *__restrict__ known_cnt,
                                                             float *__restrict__ dist2, int *__restrict__ idx) {
    __shared__ int s_unk[ST_MAXB + 1], s_kn[ST_MAXB + 1];
    build_prefix(B, unk_cnt, s_unk);
    if (threadIdx.x == 64) {
        int acc = 0;
        for (int k = 0; k < B; ++k) { s_kn[k] = acc; acc += known_cnt[k]; }
        s_kn[B] = acc;
    }
    __syncthreads();
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= N) return;
    const int bs = sample_of(pt, B, s_unk);
    const int start = s_kn[bs], m = s_kn[bs + 1] - s_kn[bs];
    const float *__restrict__ kn = known + (size_t)start * 3;
    const float ux = unknown[(size_t)pt * 3], uy = unknown[(size_t)pt * 3 + 1], uz = unknown[(size_t)pt * 3 + 2];
    // the reference keeps the bests as doubles initialised to 1e40 (interpolate_gpu.cu:49): for a float d,
    // d < 1e40 <=> d < +inf, and an unfilled slot stores (float)1e40 = +inf — so +inf floats reproduce it exactly
    const float INF = __builtin_inff();
    float b1 = INF, b2 = INF, b3 = INF;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k = 0; k < m; ++k) {
        const float d = sqdist(ux - kn[(size_t)k * 3], uy - kn[(size_t)k * 3 + 1], uz - kn[(size_t)k * 3 + 2]);
        if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
        else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
        else if (d < b3) { b3 = d; i3 = k; }
    }
    dist2[(size_t)pt * 3] = b1; dist2[(size_t)pt * 3 + 1] = b2; dist2[(size_t)pt * 3 + 2] = b3;
    idx[(size_t)pt * 3] = i1 + start; idx[(size_t)pt * 3 + 1] = i2 + start; idx[(size_t)pt * 3 + 2] = i3 + start;
}

// ---- interpolation: out (N, C) = w0 f[i0] + w1 f[i1] + w2 f[i2] (pinned fma order), lanes over channels -----
__global__ __launch_bounds__(256) void stack_three_interpolate_kernel(long long total, int C,
                                                                      const float *__restrict__ features,
                                                                      const int *__restrict__ idx,
                                                                      const float *__restrict__ weight,
                                                                      float *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long pt = e / C;
        const int c = (int)(e - pt * C);
        const float p0 = features[(size_t)idx[pt * 3] * C + c], p1 = features[(size_t)idx[pt * 3 + 1] * C + c],
                    p2 = features[(size_t)idx[pt * 3 + 2] * C + c];
        out[e] = __fmaf_rn(weight[pt * 3 + 2], p2, __fmaf_rn(weight[pt * 3 + 1], p1, __fmul_rn(weight[pt * 3], p0)));
    }
}

__global__ __launch_bounds__(256) void stack_three_interpolate_grad_kernel(long long total, int C,
                                                                           const float *__restrict__ grad_out,
                                                                           const int *__restrict__ idx,
                                                                           const float *__restrict__ weight,
                                                                           float *__restrict__ grad_features) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long pt = e / C;
        const int c = (int)(e - pt * C);
        const float g = grad_out[e];
        atomicAdd(grad_features + (size_t)idx[pt * 3] * C + c, g * weight[pt * 3]);
        atomicAdd(grad_features + (size_t)idx[pt * 3 + 1] * C + c, g * weight[pt * 3 + 1]);
        atomicAdd(grad_features + (size_t)idx[pt * 3 + 2] * C + c, g * weight[pt * 3 + 2]);
    }
}

static unsigned grid_for(long long total) {
    const long long want = (total + 255) / 256;
    return (unsigned)(want < 1 ? 1 : want > 65536 ? 65536 : want);
}

}  // namespace pdm

using namespace pdm;

extern "C" int pdm_stack_ball_query(void *stream, int B, int M, float radius, int nsample, const float *new_xyz,
                                    const int *new_xyz_batch_cnt, const float *xyz, const int *xyz_batch_cnt, int *idx) {
    PDM_REQUIRE(B >= 1 && B <= ST_MAXB && M >= 0 && nsample > 0, PDM_E_BADARG, "stack_ball_query: B=%d M=%d nsample=%d", B, M, nsample);
    if (M == 0) return 0;
    PDM_REQUIRE(new_xyz && new_xyz_batch_cnt && xyz_batch_cnt && idx, PDM_E_BADARG, "stack_ball_query: null pointer");
    hipLaunchKernelGGL(stack_ball_query_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), B, M, radius, nsample,
                       new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx);
    return check_launch("stack_ball_query");
}

extern "C" int pdm_stack_group_points(void *stream, int B, int M, int C, int nsample, const float *features,
                                      const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt, float *out) {
    PDM_REQUIRE(B >= 1 && B <= ST_MAXB && M >= 0 && C >= 0 && nsample >= 0, PDM_E_BADARG, "stack_group_points: bad size");
    const long long total = (long long)M * C * nsample;
    if (total == 0) return 0;
    PDM_REQUIRE(features && features_batch_cnt && idx && idx_batch_cnt && out, PDM_E_BADARG, "stack_group_points: null pointer");
    hipLaunchKernelGGL(stack_group_points_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), B, M, C, nsample,
                       features, features_batch_cnt, idx, idx_batch_cnt, out);
    return check_launch("stack_group_points");
}

extern "C" int pdm_stack_group_points_grad(void *stream, int B, int M, int C, int N, int nsample, const float *grad_out,
                                           const int *idx, const int *idx_batch_cnt, const int *features_batch_cnt,
                                           float *grad_features) {
    PDM_REQUIRE(B >= 1 && B <= ST_MAXB && M >= 0 && C >= 0 && N >= 0 && nsample >= 0, PDM_E_BADARG, "stack_group_points_grad: bad size");
    const long long total = (long long)M * C * nsample;
    if (total == 0) return 0;
    PDM_REQUIRE(grad_out && features_batch_cnt && idx && idx_batch_cnt && grad_features, PDM_E_BADARG, "stack_group_points_grad: null pointer");
    hipLaunchKernelGGL(stack_group_points_grad_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), B, M, C, nsample,
                       grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features);
    return check_launch("stack_group_points_grad");
}

extern "C" int pdm_stack_three_nn(void *stream, int B, int N, const float *unknown, const int *unknown_batch_cnt,
                                  const float *known, const int *known_batch_cnt, float *dist2, int *idx) {
    PDM_REQUIRE(B >= 1 && B <= ST_MAXB && N >= 0, PDM_E_BADARG, "stack_three_nn: B=%d N=%d", B, N);
    if (N == 0) return 0;
    PDM_REQUIRE(unknown && unknown_batch_cnt && known_batch_cnt && dist2 && idx, PDM_E_BADARG, "stack_three_nn: null pointer");
    hipLaunchKernelGGL(stack_three_nn_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), B, N, unknown,
                       unknown_batch_cnt, known, known_batch_cnt, dist2, idx);
    return check_launch("stack_three_nn");
}

extern "C" int pdm_stack_three_interpolate(void *stream, int N, int C, const float *features, const int *idx,
                                           const float *weight, float *out) {
    PDM_REQUIRE(N >= 0 && C >= 0, PDM_E_BADARG, "stack_three_interpolate: bad size");
    const long long total = (long long)N * C;
    if (total == 0) return 0;
    PDM_REQUIRE(features && idx && weight && out, PDM_E_BADARG, "stack_three_interpolate: null pointer");
    hipLaunchKernelGGL(stack_three_interpolate_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), total, C, features,
                       idx, weight, out);
    return check_launch("stack_three_interpolate");
}

extern "C" int pdm_stack_three_interpolate_grad(void *stream, int N, int C, const float *grad_out, const int *idx,
                                                const float *weight, float *grad_features) {
    PDM_REQUIRE(N >= 0 && C >= 0, PDM_E_BADARG, "stack_three_interpolate_grad: bad size");
    const long long total = (long long)N * C;
    if (total == 0) return 0;
    PDM_REQUIRE(grad_out && idx && weight && grad_features, PDM_E_BADARG, "stack_three_interpolate_grad: null pointer");
    hipLaunchKernelGGL(stack_three_interpolate_grad_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), total, C,
                       grad_out, idx, weight, grad_features);
    return check_launch("stack_three_interpolate_grad");
}
