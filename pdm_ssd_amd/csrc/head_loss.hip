// Point head: target assignment + both losses + their gradients in TWO launches (training).
//
// Restates, per point, what pdm_ssd_amd/dense_heads/point_head_template.py does with ~100 elementwise torch kernels
// (behaviour of /root/reference/pcdet/models/dense_heads/point_head_template.py:51-139 assign_stack_targets with
// set_ignore_flag, :141-206 get_cls_layer_loss / get_box_layer_loss, utils/box_coder_utils.py:156-179 PointResidualCoder
// .encode_torch with mean sizes, utils/loss_utils.py:10-74 SigmoidFocalClassificationLoss, :76-141 WeightedSmoothL1Loss):
//
//   label   = class of the box that holds the point | -1 inside only the enlarged box | 0 background
//   L_cls   = w_cls * sum_{points with label >= 0, classes c} focal(x_c, [label == c]) / max(#positives, 1)
//   L_box   = w_box * sum_{label > 0, codes k} smooth_l1(code_w_k (pred_k - target_k), beta) / max(#positives, 1)
//   target  = ((xg - x) / diag, (yg - y) / diag, (zg - z) / dza, log(dxg / dxa), log(dyg / dya), log(dzg / dza), cos r, sin r)
//
// Launch 1 counts the positives (and checks the classes against the mean-size table: the reference asserts there; here the box
// loss turns NaN).  Launch 2 forms labels, loss partials (one pair per workgroup, folded in double in a fixed order by the
// third, one-workgroup launch: bit-reproducible) and d L / d pred for both stacks, in the predictions' dtype.
#include "common.h"

namespace pdm {

constexpr int HL_T = 256;

struct HeadLossArgs {
    long long n_total;
    int n_per_sample, boxes_per_sample, num_class, n_mean;
    int pred_bf16;                       // 0: fp32 predictions / gradients, 1: bf16
    const void *cls_preds; long long cls_stride;      // elements between rows
    const void *box_preds; long long box_stride;
    const float *xyz; long long xyz_stride;           // xyz[i * stride + 0..2]
    const int *box_idx, *ext_idx;                     // (B n): first containing box / enlarged box, -1 = none
    const float *gt_boxes;                            // (B, M, 8) [x y z dx dy dz heading class]
    const float *mean_size;                           // (n_mean, 3)
    float code_w[8];
    float beta, alpha, gamma, cls_weight, box_weight;
    long long *labels;                                // (B n) out
    void *dcls, *dbox;                                // (B n, num_class) / (B n, 8) out, contiguous
    float *partials;                                  // (blocks, 2)
    int *counts;                                      // [0] positives, [1] 1 when a positive's class exceeds the mean-size table
    float *out;                                       // [0] L_cls, [1] L_box, [2] #positives
};

__device__ __forceinline__ float hl_load(const void *p, long long i, int bf16) {
    return bf16 ? __uint_as_float((unsigned)static_cast<const unsigned short *>(p)[i] << 16) : static_cast<const float *>(p)[i];
}
__device__ __forceinline__ unsigned short hl_bf16(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ void hl_store(void *p, long long i, float v, int bf16) {
    if (bf16) static_cast<unsigned short *>(p)[i] = hl_bf16(v);
    else static_cast<float *>(p)[i] = v;
}

__global__ __launch_bounds__(HL_T) void head_loss_count_kernel(HeadLossArgs a) {
    __shared__ int sh[2];
    if (threadIdx.x < 2) sh[threadIdx.x] = 0;
    __syncthreads();
    int pos = 0, bad = 0;
    for (long long i = (long long)blockIdx.x * HL_T + threadIdx.x; i < a.n_total; i += (long long)gridDim.x * HL_T) {
        const int bi = a.box_idx[i];
        if (bi >= 0) {
            const long long b = i / a.n_per_sample;
            const int cls = (int)(long long)a.gt_boxes[(b * a.boxes_per_sample + bi) * 8 + 7];
            const int label = a.num_class == 1 ? 1 : cls;
            pos += label > 0 ? 1 : 0;
            bad |= cls > a.n_mean ? 1 : 0;
        }
    }
    for (int off = 32; off >= 1; off >>= 1) { pos += __shfl_xor(pos, off, 64); bad |= __shfl_xor(bad, off, 64); }
    if ((threadIdx.x & 63) == 0) { if (pos) atomicAdd(&sh[0], pos); if (bad) atomicOr(&sh[1], 1); }
    __syncthreads();
    if (threadIdx.x == 0) { if (sh[0]) atomicAdd(&a.counts[0], sh[0]); if (sh[1]) atomicOr(&a.counts[1], 1); }
}

__global__ __launch_bounds__(HL_T) void head_loss_main_kernel(HeadLossArgs a) {
    __shared__ float red[2][HL_T / 64];
    const long long i = (long long)blockIdx.x * HL_T + threadIdx.x;
    float lc = 0.f, lb = 0.f;
    if (i < a.n_total) {
        const float inv_pos = 1.0f / fmaxf((float)a.counts[0], 1.0f);
        const int bi = a.box_idx[i], ei = a.ext_idx[i];
        const bool fg = bi >= 0;
        const long long b = i / a.n_per_sample;
        const float *__restrict__ gb = a.gt_boxes + (b * a.boxes_per_sample + (fg ? bi : 0)) * 8;
        const int cls = (int)(long long)gb[7];
        int label = (fg != (ei >= 0)) ? -1 : 0;          // inside exactly one of (box, enlarged box): ignored
        if (fg) label = a.num_class == 1 ? 1 : cls;
        a.labels[i] = label;
        // ---- classification: sigmoid focal loss over the classes, weight 1 / #positives for every non-ignored point
        const float wc = label >= 0 ? inv_pos : 0.f;
        for (int c = 1; c <= a.num_class; ++c) {
            const float x = hl_load(a.cls_preds, i * a.cls_stride + (c - 1), a.pred_bf16);
            const float t = label == c ? 1.f : 0.f;
            const float p = 1.0f / (1.0f + expf(-x));
            const float miss = p + t * ((1.0f - p) - p);                         // torch.lerp(p, 1 - p, t)
            const float balance = (1.0f - a.alpha) + t * (2.0f * a.alpha - 1.0f);
            const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));    // binary_cross_entropy_with_logits
            const float mg = a.gamma == 2.0f ? miss * miss : powf(miss, a.gamma);
            lc += balance * mg * bce * wc;
            const float dmg = a.gamma == 2.0f ? 2.0f * miss : a.gamma * powf(miss, a.gamma - 1.0f);
            const float dper = balance * (dmg * (1.0f - 2.0f * t) * p * (1.0f - p) * bce + mg * (p - t));
            hl_store(a.dcls, i * a.num_class + (c - 1), dper * wc * a.cls_weight, a.pred_bf16);
        }
        // ---- box regression: foreground points only
        if (label > 0) {
            int ac = cls < 1 ? 1 : cls > a.n_mean ? a.n_mean : cls;                 // gt_classes.clamp(1, n_mean)
            const float dxa = a.mean_size[(ac - 1) * 3], dya = a.mean_size[(ac - 1) * 3 + 1], dza = a.mean_size[(ac - 1) * 3 + 2];
            const float diag = sqrtf(dxa * dxa + dya * dya);
            const float px = a.xyz[i * a.xyz_stride], py = a.xyz[i * a.xyz_stride + 1], pz = a.xyz[i * a.xyz_stride + 2];
            const float dxg = fmaxf(gb[3], 1e-5f), dyg = fmaxf(gb[4], 1e-5f), dzg = fmaxf(gb[5], 1e-5f);
            float tgt[8];
            tgt[0] = (gb[0] - px) / diag; tgt[1] = (gb[1] - py) / diag; tgt[2] = (gb[2] - pz) / dza;
            tgt[3] = logf(dxg / dxa); tgt[4] = logf(dyg / dya); tgt[5] = logf(dzg / dza);
            tgt[6] = cosf(gb[6]); tgt[7] = sinf(gb[6]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float x = hl_load(a.box_preds, i * a.box_stride + k, a.pred_bf16);
                const float r = (x - tgt[k]) * a.code_w[k];
                const float mag = fabsf(r);
                float l, d;
                if (a.beta < 1e-5f) { l = mag; d = r > 0.f ? 1.f : r < 0.f ? -1.f : 0.f; }
                else if (mag < a.beta) { l = mag * mag * (0.5f / a.beta); d = r / a.beta; }
                else { l = mag - 0.5f * a.beta; d = r > 0.f ? 1.f : -1.f; }
                lb += l * inv_pos;
                hl_store(a.dbox, i * 8 + k, d * a.code_w[k] * inv_pos * a.box_weight, a.pred_bf16);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) hl_store(a.dbox, i * 8 + k, 0.f, a.pred_bf16);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) { lc += __shfl_xor(lc, off, 64); lb += __shfl_xor(lb, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lc; red[1][threadIdx.x >> 6] = lb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.partials[(size_t)blockIdx.x * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        a.partials[(size_t)blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(HL_T) void head_loss_finalize_kernel(HeadLossArgs a, int nblocks) {
    __shared__ double sh[2][HL_T];
    double sc = 0.0, sb = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += HL_T) { sc += a.partials[(size_t)k * 2]; sb += a.partials[(size_t)k * 2 + 1]; }
    sh[0][threadIdx.x] = sc; sh[1][threadIdx.x] = sb;
    __syncthreads();
    for (int half = HL_T / 2; half >= 1; half >>= 1) {
        if ((int)threadIdx.x < half) { sh[0][threadIdx.x] += sh[0][threadIdx.x + half]; sh[1][threadIdx.x] += sh[1][threadIdx.x + half]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.out[0] = (float)sh[0][0] * a.cls_weight;
        a.out[1] = a.counts[1] ? __builtin_nanf("") : (float)sh[1][0] * a.box_weight;
        a.out[2] = (float)a.counts[0];
    }
}

}  // namespace pdm

using namespace pdm;

extern "C" size_t pdm_point_head_loss_workspace_bytes(long long n_total) {
    const long long blocks = (n_total + HL_T - 1) / HL_T;
    return (size_t)(16 + blocks * 2 * (long long)sizeof(float));
}

// See the file header.  n_total = B * n_per_sample points in sample order; box_idx / ext_idx from pdm_points_in_boxes on the boxes
// and the enlarged boxes; pred_bf16: dtype of cls_preds / box_preds AND of the gradients dcls (n_total, num_class) / dbox
// (n_total, 8).  labels (n_total) int64.  out (3) fp32: [L_cls, L_box (NaN when a positive's class exceeds the mean-size table),
// #positives].  workspace: pdm_point_head_loss_workspace_bytes(n_total) bytes, 16-byte aligned.
extern "C" int pdm_point_head_loss(void *stream, long long n_total, int n_per_sample, int boxes_per_sample, int num_class, int n_mean,
                                   int pred_bf16, const void *cls_preds, long long cls_stride, const void *box_preds, long long box_stride,
                                   const float *xyz, long long xyz_stride, const int *box_idx, const int *ext_idx, const float *gt_boxes,
                                   const float *mean_size, const float *code_weights, float beta, float alpha, float gamma, float cls_weight,
                                   float box_weight, long long *labels, void *dcls, void *dbox, float *out, void *workspace,
                                   size_t workspace_bytes) {
    PDM_REQUIRE(n_total >= 0 && n_per_sample >= 1 && boxes_per_sample >= 1 && num_class >= 1 && n_mean >= 1, PDM_E_BADARG,
                "point_head_loss: bad size");
    PDM_REQUIRE(out && workspace && workspace_bytes >= pdm_point_head_loss_workspace_bytes(n_total), PDM_E_BADARG,
                "point_head_loss: null output or workspace too small");
    PDM_REQUIRE(n_total == 0 || (cls_preds && box_preds && xyz && box_idx && ext_idx && gt_boxes && mean_size && code_weights && labels &&
                                 dcls && dbox), PDM_E_BADARG, "point_head_loss: null pointer");
    PDM_REQUIRE(n_total % n_per_sample == 0, PDM_E_BADARG, "point_head_loss: n_total is not a multiple of n_per_sample");
    HeadLossArgs a{};
    a.n_total = n_total; a.n_per_sample = n_per_sample; a.boxes_per_sample = boxes_per_sample; a.num_class = num_class; a.n_mean = n_mean;
    a.pred_bf16 = pred_bf16 ? 1 : 0;
    a.cls_preds = cls_preds; a.cls_stride = cls_stride; a.box_preds = box_preds; a.box_stride = box_stride;
    a.xyz = xyz; a.xyz_stride = xyz_stride; a.box_idx = box_idx; a.ext_idx = ext_idx; a.gt_boxes = gt_boxes; a.mean_size = mean_size;
    for (int k = 0; k < 8; ++k) a.code_w[k] = code_weights[k];     // host array
    a.beta = beta; a.alpha = alpha; a.gamma = gamma; a.cls_weight = cls_weight; a.box_weight = box_weight;
    a.labels = labels; a.dcls = dcls; a.dbox = dbox; a.out = out;
    a.counts = static_cast<int *>(workspace);
    a.partials = reinterpret_cast<float *>(static_cast<char *>(workspace) + 16);
    hipError_t e = hipMemsetAsync(workspace, 0, 16, as_stream(stream));
    PDM_REQUIRE(e == hipSuccess, PDM_E_BADARG, "point_head_loss: memset failed");
    const long long blocks = (n_total + HL_T - 1) / HL_T;
    if (blocks > 0) {
        const int cgrid = (int)(blocks < 1024 ? blocks : 1024);
        hipLaunchKernelGGL(head_loss_count_kernel, dim3(cgrid), dim3(HL_T), 0, as_stream(stream), a);
        hipLaunchKernelGGL(head_loss_main_kernel, dim3((unsigned)blocks), dim3(HL_T), 0, as_stream(stream), a);
    }
    hipLaunchKernelGGL(head_loss_finalize_kernel, dim3(1), dim3(HL_T), 0, as_stream(stream), a, (int)blocks);
    return check_launch("point_head_loss");
}
