// Heat-map head (the dense half of the hybrid head) in training: gaussian targets, penalty-reduced focal loss and its gradient
// in FOUR launches instead of ~75 elementwise torch kernels per step.
//
// Restates pdm_ssd_amd/dense_heads/pdm_heatmap_head.py::assign_targets + utils/centernet_utils.py (behaviour of
// /root/reference/pcdet/models/model_utils/centernet_utils.py:9-70 gaussian_radius / gaussian2D / draw_gaussian_to_heatmap as
// /root/reference/pcdet/models/dense_heads/center_head.py:100-160 calls them) and utils/loss_utils.py::neg_loss_cornernet on
// clamp(sigmoid(logits), 1e-4, 1 - 1e-4) (center_head.py:232, /root/reference/pcdet/utils/loss_utils.py:266-304):
//
//   target[b, c, y, x] = max over the boxes of (b, c) of exp(-(dx^2 + dy^2) / (2 sigma^2)) inside the box's (2 r + 1)^2 window,
//                        r = max(int(gaussian_radius(dy_cells, dx_cells, overlap)), min_radius), sigma = (2 r + 1) / 6,
//                        values under fp32 eps dropped
//   S      = sum_{target == 1} (1 - p)^2 log p + sum_{target < 1} (1 - target)^4 p^2 log(1 - p),   p = clamp(sigmoid(x))
//   L      = - weight * S / max(#{target == 1}, 1)
//
// Launch 1 (after a memset) draws the gaussians with integer atomic max (the values are non-negative floats: their bit patterns
// order like the values).  Launch 2 forms per-workgroup partial sums of the two terms and the peak count, and d S / d x per
// element; launch 3 (one workgroup) folds the partials in double in a fixed order (bit-reproducible), leaving L and the factor
// - weight / max(#peaks, 1) that backward() multiplies the stored d S / d x by.
#include "common.h"

namespace pdm {

constexpr int HM_T = 256;

// centernet_utils.gaussian_radius in torch's fp32 evaluation order: the python scalars (1 - o, 1 + o, -2 o, o - 1, 4 (4 o)) are
// formed in double and enter the tensor arithmetic as fp32 factors
__device__ __forceinline__ float hm_gaussian_radius(float height, float width, double o) {
    const float k1m = (float)(1.0 - o), k1p = (float)(1.0 + o), kn2 = (float)(-2.0 * o), km1 = (float)(o - 1.0), k16 = (float)(4.0 * (4.0 * o));
    const float b1 = height + width;
    const float c1 = __fmul_rn(__fmul_rn(__fmul_rn(width, height), k1m), __fdiv_rn(1.0f, k1p));   // (tensor / python scalar = tensor * (1 / scalar) in torch's kernel)
    const float r1 = __fmul_rn(__fadd_rn(b1, __fsqrt_rn(__fsub_rn(__fmul_rn(b1, b1), __fmul_rn(4.0f, c1)))), 0.5f);
    const float b2 = __fmul_rn(2.0f, height + width);
    const float c2 = __fmul_rn(__fmul_rn(k1m, width), height);
    const float r2 = __fmul_rn(__fadd_rn(b2, __fsqrt_rn(__fsub_rn(__fmul_rn(b2, b2), __fmul_rn(16.0f, c2)))), 0.5f);
    const float b3 = __fmul_rn(kn2, height + width);
    const float c3 = __fmul_rn(__fmul_rn(km1, width), height);
    const float r3 = __fmul_rn(__fadd_rn(b3, __fsqrt_rn(__fsub_rn(__fmul_rn(b3, b3), __fmul_rn(k16, c3)))), 0.5f);
    return fminf(fminf(r1, r2), r3);
}

struct HmTargetArgs {
    int B, M, C, H, W;
    const float *gt_boxes;          // (B, M, 8) [x y z dx dy dz heading class], zero rows = padding
    float x0, y0, vx, vy, stride;   // point-cloud range minimum, voxel size, feature-map stride: cell = (x - x0) / vx / stride
    double min_overlap;
    int min_radius, max_radius;
    float *heatmap;                 // (B, C, H, W), zero on entry
};

// one workgroup per box; thread = cell of the (2 max_radius + 1)^2 window
__global__ __launch_bounds__(HM_T) void hm_target_kernel(HmTargetArgs a) {
    const int bm = blockIdx.x, b = bm / a.M;
    const float *g = a.gt_boxes + (size_t)bm * 8;
    const float cls = g[7];
    // (division of a tensor by a python scalar is a multiplication by the scalar's fp32 reciprocal in torch's kernel: the same here,
    //  so that the integer cell and radius of a box come out the same)
    const float ivx = __fdiv_rn(1.0f, a.vx), ivy = __fdiv_rn(1.0f, a.vy), is = __fdiv_rn(1.0f, a.stride);
    const float dxc = __fmul_rn(__fmul_rn(g[3], ivx), is), dyc = __fmul_rn(__fmul_rn(g[4], ivy), is);
    if (!(dxc > 0.0f) || !(dyc > 0.0f) || !(cls >= 1.0f)) return;   // padding / degenerate box (uniform over the workgroup)
    const int c = (int)cls - 1;
    if (c >= a.C) return;
    const float cx = fminf(fmaxf(__fmul_rn(__fmul_rn(g[0] - a.x0, ivx), is), 0.0f), (float)a.W - 0.5f);
    const float cy = fminf(fmaxf(__fmul_rn(__fmul_rn(g[1] - a.y0, ivy), is), 0.0f), (float)a.H - 0.5f);
    const int ix = (int)cx, iy = (int)cy;
    int r = (int)hm_gaussian_radius(dxc, dyc, a.min_overlap);     // (height, width) = (dx, dy) cells as the head passes them
    if (r < a.min_radius) r = a.min_radius;
    const float rt = (float)r;
    const int rw = r < a.max_radius ? r : a.max_radius;          // the window is clipped at max_radius, sigma keeps the true radius
    const float sigma = __fmul_rn(__fadd_rn(__fmul_rn(2.0f, rt), 1.0f), __fdiv_rn(1.0f, 6.0f));
    const float den = __fmul_rn(__fmul_rn(2.0f, sigma), sigma);
    const int K = 2 * a.max_radius + 1;
    int *map = reinterpret_cast<int *>(a.heatmap + ((size_t)b * a.C + c) * a.H * a.W);
    for (int e = threadIdx.x; e < K * K; e += HM_T) {
        const int dy = e / K - a.max_radius, dx = e % K - a.max_radius;
        if (abs(dx) > rw || abs(dy) > rw) continue;
        const int x = ix + dx, y = iy + dy;
        if (x < 0 || x >= a.W || y < 0 || y >= a.H) continue;
        const float v = expf(-__fdiv_rn((float)(dx * dx + dy * dy), den));
        if (v < 1.1920928955078125e-07f) continue;               // h[h < eps * h.max()] = 0 (the window's maximum is 1)
        atomicMax(map + (size_t)y * a.W + x, __float_as_int(v));
    }
}

struct HmLossArgs {
    long long n;                    // B C H W
    int C, H, W;
    const void *logits; int logits_bf16;
    long long sb, sc, sh, sw;       // element strides of the logits
    const float *heatmap;           // (B, C, H, W) contiguous
    float *dlogits;                 // (B, C, H, W) contiguous fp32: d S / d logit
    double *partials;               // (blocks, 3): peak term, other term, peaks
    float weight;
    float *out;                     // [0] loss, [1] - weight / max(peaks, 1), [2] peaks
};

__global__ __launch_bounds__(HM_T) void hm_loss_kernel(HmLossArgs a) {
    __shared__ double red[3][HM_T / 64];
    double s_peak = 0.0, s_else = 0.0, n_peak = 0.0;
    const long long hw = (long long)a.H * a.W, chw = hw * a.C;
    for (long long i = (long long)blockIdx.x * HM_T + threadIdx.x; i < a.n; i += (long long)gridDim.x * HM_T) {
        const long long b = i / chw, r0 = i - b * chw, c = r0 / hw, r1 = r0 - c * hw, y = r1 / a.W, x = r1 - y * a.W;
        const long long li = b * a.sb + c * a.sc + y * a.sh + x * a.sw;
        const float xl = a.logits_bf16 ? __uint_as_float((unsigned)static_cast<const unsigned short *>(a.logits)[li] << 16)
                                       : static_cast<const float *>(a.logits)[li];
        const float gt = a.heatmap[i];
        const float praw = __fdiv_rn(1.0f, 1.0f + expf(-xl));
        const float lo = 1e-4f, hi = 1.0f - 1e-4f;
        const float p = fminf(fmaxf(praw, lo), hi);
        const float dp = (praw >= lo && praw <= hi) ? praw * (1.0f - praw) : 0.0f;      // clamp passes the gradient inside [lo, hi]
        const float q = 1.0f - p;
        float ds;
        if (gt == 1.0f) {
            const float lp = logf(p);
            s_peak += (double)(q * q * lp);
            n_peak += 1.0;
            ds = -2.0f * q * lp + q * q / p;
        } else {
            const float w4 = powf(1.0f - gt, 4.0f), lq = logf(q);
            s_else += (double)(w4 * (p * p) * lq);
            ds = w4 * (2.0f * p * lq - p * p / q);
        }
        a.dlogits[i] = ds * dp;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        s_peak += __shfl_xor(s_peak, off, 64); s_else += __shfl_xor(s_else, off, 64); n_peak += __shfl_xor(n_peak, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s_peak; red[1][wave] = s_else; red[2][wave] = n_peak; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int w = 0; w < HM_T / 64; ++w) s += red[threadIdx.x][w];
        a.partials[(size_t)blockIdx.x * 3 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(64) void hm_loss_finalize_kernel(int blocks, const double *__restrict__ partials, float weight, float *__restrict__ out) {
    // lane l adds partials l, l + 64, ... in order, then a butterfly over the wave: a fixed order (bit-reproducible) without the
    // 2048-step dependent chain one thread walked (180 us of a training step)
    double s_peak = 0.0, s_else = 0.0, n_peak = 0.0;
    for (int k = threadIdx.x; k < blocks; k += 64) { s_peak += partials[3 * k]; s_else += partials[3 * k + 1]; n_peak += partials[3 * k + 2]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s_peak += __shfl_xor(s_peak, off, 64); s_else += __shfl_xor(s_else, off, 64); n_peak += __shfl_xor(n_peak, off, 64);
    }
    if (threadIdx.x != 0) return;
    const double den = n_peak > 1.0 ? n_peak : 1.0;
    out[0] = (float)(-(s_peak + s_else) / den * (double)weight);
    out[1] = (float)(-(double)weight / den);
    out[2] = (float)n_peak;
}

static inline int hm_blocks(long long n) {
    const long long want = (n + HM_T * 8 - 1) / (HM_T * 8);
    return (int)(want < 1 ? 1 : want > 2048 ? 2048 : want);
}

}  // namespace pdm

using namespace pdm;

// heatmap (B, C, H, W) fp32, fully written (zeroed here, then max-merged gaussians).  gt_boxes (B, M, 8) fp32.
extern "C" int pdm_heatmap_targets(void *stream, int B, int M, int C, int H, int W, const float *gt_boxes, float x0, float y0, float vx,
                                   float vy, float stride, double min_overlap, int min_radius, int max_radius, float *heatmap) {
    PDM_REQUIRE(B >= 0 && M >= 0 && C >= 1 && H >= 1 && W >= 1 && max_radius >= 0 && max_radius <= 64 && vx > 0.0f && vy > 0.0f && stride > 0.0f,
                PDM_E_BADARG, "heatmap_targets: bad size");
    if (B == 0) return 0;
    PDM_REQUIRE(heatmap && (M == 0 || gt_boxes), PDM_E_BADARG, "heatmap_targets: null pointer");
    const hipError_t e = hipMemsetAsync(heatmap, 0, sizeof(float) * (size_t)B * C * H * W, as_stream(stream));
    PDM_REQUIRE(e == hipSuccess, PDM_E_BADARG, "heatmap_targets: memset failed");
    if (M == 0) return 0;
    PDM_REQUIRE((long long)B * M <= 0x7fffffffll, PDM_E_TOOLARGE, "heatmap_targets: %lld boxes", (long long)B * M);
    HmTargetArgs a{B, M, C, H, W, gt_boxes, x0, y0, vx, vy, stride, min_overlap, min_radius, max_radius, heatmap};
    hipLaunchKernelGGL(hm_target_kernel, dim3((unsigned)(B * M)), dim3(HM_T), 0, as_stream(stream), a);
    return check_launch("heatmap_targets");
}

extern "C" size_t pdm_heatmap_focal_loss_workspace_bytes(long long n) { return n <= 0 ? 0 : (size_t)hm_blocks(n) * 3 * sizeof(double); }

// logits (B, C, H, W) fp32 or bf16 with element strides (sb, sc, sh, sw); heatmap and dlogits (B, C, H, W) contiguous fp32;
// out[0] = loss, out[1] = the factor backward multiplies dlogits by, out[2] = number of peaks.
extern "C" int pdm_heatmap_focal_loss(void *stream, int B, int C, int H, int W, const void *logits, int logits_bf16, long long sb, long long sc,
                                      long long sh, long long sw, const float *heatmap, float weight, float *dlogits, float *out,
                                      void *workspace, size_t workspace_bytes) {
    PDM_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 1, PDM_E_BADARG, "heatmap_focal_loss: bad size");
    PDM_REQUIRE(out, PDM_E_BADARG, "heatmap_focal_loss: null pointer");
    const long long n = (long long)B * C * H * W;
    const int blocks = hm_blocks(n);
    if (n > 0) {
        PDM_REQUIRE(logits && heatmap && dlogits && workspace, PDM_E_BADARG, "heatmap_focal_loss: null pointer");
        PDM_REQUIRE(workspace_bytes >= pdm_heatmap_focal_loss_workspace_bytes(n) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                    PDM_E_BADARG, "heatmap_focal_loss: workspace of %zu bytes, need %zu (8-byte aligned)", workspace_bytes,
                    pdm_heatmap_focal_loss_workspace_bytes(n));
        HmLossArgs a{n, C, H, W, logits, logits_bf16, sb, sc, sh, sw, heatmap, dlogits, static_cast<double *>(workspace), weight, out};
        hipLaunchKernelGGL(hm_loss_kernel, dim3((unsigned)blocks), dim3(HM_T), 0, as_stream(stream), a);
        if (int rc = check_launch("heatmap_focal_loss")) return rc;
    }
    hipLaunchKernelGGL(hm_loss_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), n > 0 ? blocks : 0, static_cast<const double *>(workspace),
                       weight, out);
    return check_launch("heatmap_focal_loss(finalize)");
}
