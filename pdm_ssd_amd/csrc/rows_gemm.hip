// LDS-tiled fp32 MFMA GEMM for ONE layer over plain point-major rows:
//     out[r][c] = act(sum_k in[r][k] * W[c][k] + bias[c]),      act = ReLU or identity,
// the shape of the hoisted first-layer blocks (z = W1[:, wide] f, fused_mlp.hip section "pre") and of 1x1
// convolutions over rows.  The fused chain kernels give every 16/32 positions their own pass over the weights
// (L2 -> L1 weight streaming is what bounds them on wide layers); here a workgroup owns 128 rows x 128 channels,
// stages both operands through LDS once per 16-deep k-block and shares them among its 8 waves:
//   LDS stage = weights: 8 blocks x 64 lanes x float4 exactly as packed on the host (fragment order, so the
//               A fragment of a wave is one conflict-free ds_read_b128 per block);
//               rows:    128 x (16 + 4 pad) floats (B fragment = one ds_read_b128 per 16-row tile);
//   wave (wp, wc) = rows [64 wp, +64) x channel blocks [2 wc, +2): 8 accumulator tiles, 6 fragment reads per
//               32 MFMAs; two stages, the global loads of k-block kb+2 are in flight while kb computes.
// Same transposed contraction and packed-weight format as fused_mlp.hip (D layout = 4 consecutive channels of
// one row per lane -> 16-byte row stores).
#include "common.h"

namespace pdm {

typedef float gf4 __attribute__((ext_vector_type(4)));

constexpr int RG_ROWS = 128;   // rows per workgroup
constexpr int RG_BLKS = 8;     // 16-channel blocks per workgroup (128 channels)
constexpr int RG_HS = 24;      // LDS row stride of the activation stage in floats (16 + 8 pad): with 6 slots of 16 B per row the 16 lanes of every
                               // ds_read_b128 group ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) land on 16 different slots of the 256-B bank row;
                               // with 5 (a 4-float pad) rows p and p + 4 of neighbouring k-quads shared a slot: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.375

struct RowsGemmArgs {
    int rows, cin, in_stride;   // in (rows, in_stride), cin real input channels
    int nkb, nmb;               // padded K / 16, padded C / 16
    int cout, out_stride;
    const float *in;
    const float *wpack;         // [mb][kb][lane][4]
    const float *bias;          // padded
    float *out;
    float floor;                // 0 = ReLU, -inf = linear
    // FP_PRE form (two-layer FP module with a tiny skip input, hoisted known features): the rows this GEMM
    // contracts are layer-1 activations made on the fly while staging,
    //   h1[r][c] = relu(sum_k w_k z[idx_k][c] + sum_s W1s[c][s] skip[r][s] + b1[c])      (fused_mlp.hip "pre")
    int n, m, z_stride, c_skip; // rows = b * n fine points, m known points per sample
    const float *z;             // (b, m, z_stride)
    const float *skip;          // (rows, c_skip) or null
    const int *idx;             // (rows, 3)
    const float *weight;        // (rows, 3)
    const float *w1;            // layer-1 packed weights [mb][1][lane][4] (only k < c_skip is non-zero)
    const float *b1;            // layer-1 bias, padded
};

template <bool FP_PRE>
__global__ __launch_bounds__(512) void rows_gemm_kernel(RowsGemmArgs a) {
    __shared__ __attribute__((aligned(16))) float s_w[2][RG_BLKS * 64 * 4];
    __shared__ __attribute__((aligned(16))) float s_h[2][RG_ROWS * RG_HS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4;
    const int wp = wave >> 2, wc = wave & 3;
    // FP_PRE: every row gathers three z rows of ITS cloud's known set (m x z_stride floats: 2 MB for FP1 of PointNet2MSG).
    // Workgroups go to the 8 XCDs round-robin by linear id, so with the plain tile order each XCD's 4 MB L2 sees the
    // z rows of every cloud (64 MB at bs=32) and keeps none.  When the tiles divide evenly, XCD x works through clouds
    // x, x+8, x+16, ... instead: its L2 then holds the one or two known sets in flight (PMC, FP1 at bs=32: HBM reads
    // 442 -> 140 MB per launch; the launch itself stays at 0.26 ms, it was not waiting for HBM).
    int tile = blockIdx.x;
    if constexpr (FP_PRE) {
        const int tpc = a.n / RG_ROWS, nb = a.rows / a.n;          // tiles per cloud, clouds
        if (gridDim.y == 1 && tpc * RG_ROWS == a.n && (nb & 7) == 0) {
            const int xcd = tile & 7, slot = tile >> 3;
            tile = ((slot / tpc) * 8 + xcd) * tpc + slot % tpc;
        }
    }
    const long long r0 = (long long)tile * RG_ROWS;
    const int mb0 = blockIdx.y * RG_BLKS;

    // staging roles: this thread loads float4 `lane` of weight block mb0 + wave, and channels [4 part, +4) of row hrow
    const int hrow = tid >> 2, part = tid & 3;
    const bool w_live = mb0 + wave < a.nmb;
    const gf4 *wsrc = reinterpret_cast<const gf4 *>(a.wpack) + ((size_t)(mb0 + wave) * a.nkb) * 64 + lane;
    const bool h_live = r0 + hrow < a.rows;
    const float *hsrc = a.in + (size_t)(r0 + hrow) * a.in_stride + 4 * part;
    const bool vec = (a.in_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0;

    auto load_w = [&](int kb) {
        gf4 v = {0.f, 0.f, 0.f, 0.f};
        if (w_live) v = wsrc[(size_t)kb * 64];
        return v;
    };
    // FP_PRE: this thread's row keeps its three neighbours, weights and skip values for the whole kernel
    const float *zb = nullptr;
    int o0 = 0, o1 = 0, o2 = 0;
    float iw0 = 0.f, iw1 = 0.f, iw2 = 0.f, sk[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FP_PRE) {
        if (h_live) {
            const long long row = r0 + hrow;
            const int b = (int)(row / a.n);
            zb = a.z + (size_t)b * a.m * a.z_stride + 4 * part;
            o0 = a.idx[row * 3] * a.z_stride; o1 = a.idx[row * 3 + 1] * a.z_stride; o2 = a.idx[row * 3 + 2] * a.z_stride;
            iw0 = a.weight[row * 3]; iw1 = a.weight[row * 3 + 1]; iw2 = a.weight[row * 3 + 2];
            for (int s_ = 0; s_ < a.c_skip; ++s_) sk[s_] = a.skip[row * a.c_skip + s_];
        }
    }
    // FP_PRE: the three z rows are REQUESTED here and turned into h1 only when the stage is written, one k-step
    // later, so the gathers have a whole step of MFMAs to land in
    struct Raw { gf4 za, zc, zd; };
    auto load_raw = [&](int kb) {
        Raw r;
        r.za = r.zc = r.zd = gf4{0.f, 0.f, 0.f, 0.f};
        if (h_live) {   // z width >= 16 nkb (host check)
            r.za = *reinterpret_cast<const gf4 *>(zb + o0 + 16 * kb);
            r.zc = *reinterpret_cast<const gf4 *>(zb + o1 + 16 * kb);
            r.zd = *reinterpret_cast<const gf4 *>(zb + o2 + 16 * kb);
        }
        return r;
    };
    auto finish = [&](const Raw &r, int kb) {
        gf4 v = {0.f, 0.f, 0.f, 0.f};
        if (!h_live) return v;
        const int c = 16 * kb + 4 * part;
        const gf4 bb = *reinterpret_cast<const gf4 *>(a.b1 + c);
        float h[4] = {bb.x, bb.y, bb.z, bb.w};
        const float zz[3][4] = {{r.za.x, r.za.y, r.za.z, r.za.w}, {r.zc.x, r.zc.y, r.zc.z, r.zc.w}, {r.zd.x, r.zd.y, r.zd.z, r.zd.w}};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] += __fmaf_rn(iw2, zz[2][j], __fmaf_rn(iw1, zz[1][j], __fmul_rn(iw0, zz[0][j])));   // pinned interpolation order
            const float *wr = a.w1 + ((size_t)(c >> 4) * 64 + ((c & 15) + j)) * 4;               // W1[c + j][0..3]
            for (int s_ = 0; s_ < a.c_skip; ++s_) h[j] = fmaf(wr[s_], sk[s_], h[j]);
            h[j] = fmaxf(h[j], 0.0f);
        }
        return gf4{h[0], h[1], h[2], h[3]};
    };
    auto load_h = [&](int kb) {
        gf4 v = {0.f, 0.f, 0.f, 0.f};
        if (!h_live) return v;
        const int c = 16 * kb + 4 * part;
        if (vec && c + 4 <= a.cin) return *reinterpret_cast<const gf4 *>(hsrc + 16 * kb);
        const float *p = hsrc + 16 * kb;
        if (c < a.cin) v.x = p[0];
        if (c + 1 < a.cin) v.y = p[1];
        if (c + 2 < a.cin) v.z = p[2];
        if (c + 3 < a.cin) v.w = p[3];
        return v;
    };
    auto stage = [&](int buf, gf4 w, gf4 h) {
        reinterpret_cast<gf4 *>(s_w[buf])[tid] = w;
        *reinterpret_cast<gf4 *>(&s_h[buf][hrow * RG_HS + 4 * part]) = h;
    };

    gf4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        gf4 bi = {0.f, 0.f, 0.f, 0.f};
        if (mb0 + 2 * wc + i < a.nmb) bi = *reinterpret_cast<const gf4 *>(a.bias + 16 * (mb0 + 2 * wc + i) + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = bi;
    }

    gf4 gw = load_w(0), gh = {0.f, 0.f, 0.f, 0.f};
    Raw gr;
    if constexpr (FP_PRE) { gr = load_raw(0); gh = finish(gr, 0); } else gh = load_h(0);
    stage(0, gw, gh);
    __syncthreads();
    if (a.nkb > 1) {
        gw = load_w(1);
        if constexpr (FP_PRE) gr = load_raw(1); else gh = load_h(1);
    }
    for (int kb = 0; kb < a.nkb; ++kb) {
        const int cur = kb & 1;
        gf4 fa[2], fb[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = reinterpret_cast<const gf4 *>(s_w[cur])[(2 * wc + i) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = *reinterpret_cast<const gf4 *>(&s_h[cur][((4 * wp + t) * 16 + pos) * RG_HS + 4 * g]);
        if (kb + 1 < a.nkb) {                           // that stage was last read in step kb-1, behind the barrier
            if constexpr (FP_PRE) gh = finish(gr, kb + 1);
            stage(cur ^ 1, gw, gh);
        }
        if (kb + 2 < a.nkb) {
            gw = load_w(kb + 2);
            if constexpr (FP_PRE) gr = load_raw(kb + 2); else gh = load_h(kb + 2);
        }
#define PDM_RG_MFMA(C)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int t = 0; t < 4; ++t)                 \
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].C, fb[t].C, acc[i][t], 0, 0, 0);
        PDM_RG_MFMA(x) PDM_RG_MFMA(y) PDM_RG_MFMA(z) PDM_RG_MFMA(w)
#undef PDM_RG_MFMA
        __syncthreads();
    }

#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const long long row = r0 + (4 * wp + t) * 16 + pos;
        if (row >= a.rows) continue;
        float *orow = a.out + (size_t)row * a.out_stride;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c0 = 16 * (mb0 + 2 * wc + i) + 4 * g;
            gf4 v = acc[i][t];
            v.x = __builtin_amdgcn_fmed3f(v.x, a.floor, __builtin_inff()); v.y = __builtin_amdgcn_fmed3f(v.y, a.floor, __builtin_inff());
            v.z = __builtin_amdgcn_fmed3f(v.z, a.floor, __builtin_inff()); v.w = __builtin_amdgcn_fmed3f(v.w, a.floor, __builtin_inff());
            if (c0 + 4 <= a.cout) {
                *reinterpret_cast<gf4 *>(orow + c0) = v;
            } else {
                if (c0 < a.cout) orow[c0] = v.x;
                if (c0 + 1 < a.cout) orow[c0 + 1] = v.y;
                if (c0 + 2 < a.cout) orow[c0 + 2] = v.z;
            }
        }
    }
}

// Called by pdm_rows_mlp_fused for single-layer problems large enough to fill 128 x 128 tiles.
int rows_gemm_launch(void *stream, int rows, int cin, const float *in_pm, int k0, int c1, const float *wpack,
                     const float *bias, int relu_last, float *out_pm, int out_stride, int cout) {
    RowsGemmArgs a;
    a.rows = rows; a.cin = cin; a.in_stride = cin;
    a.nkb = k0 >> 4; a.nmb = c1 >> 4;
    a.cout = cout; a.out_stride = out_stride;
    a.in = in_pm; a.wpack = wpack; a.bias = bias; a.out = out_pm;
    a.floor = relu_last ? 0.0f : -__builtin_inff();
    a.n = a.m = a.z_stride = a.c_skip = 0;
    a.z = a.skip = a.weight = a.w1 = a.b1 = nullptr; a.idx = nullptr;
    const dim3 grid((unsigned)((rows + RG_ROWS - 1) / RG_ROWS), (unsigned)((a.nmb + RG_BLKS - 1) / RG_BLKS));
    hipLaunchKernelGGL(rows_gemm_kernel<false>, grid, dim3(512), 0, as_stream(stream), a);
    return check_launch("rows_mlp_fused(gemm)");
}

// Two-layer FP module in the hoisted form with c_skip <= 4: layer 1 is made while staging, layer 2 is the GEMM.
// dims = {16, K1, C2}; wpack/bias hold both layers in the usual packed order.
int fp_pre_gemm_launch(void *stream, int b, int n, int m, int c_skip, const float *z_pm, int z_stride,
                       const float *skip_pm, const int *idx, const float *weight, int k1, int c2,
                       const float *wpack, const float *bias, float *out_pm, int out_stride, int cout) {
    RowsGemmArgs a;
    a.rows = b * n; a.cin = k1; a.in_stride = 0;
    a.nkb = k1 >> 4; a.nmb = c2 >> 4;
    a.cout = cout; a.out_stride = out_stride;
    a.in = nullptr; a.wpack = wpack + 16 * k1; a.bias = bias + k1; a.out = out_pm;
    a.floor = 0.0f;
    a.n = n; a.m = m; a.z_stride = z_stride; a.c_skip = c_skip;
    a.z = z_pm; a.skip = skip_pm; a.idx = idx; a.weight = weight; a.w1 = wpack; a.b1 = bias;
    const dim3 grid((unsigned)((a.rows + RG_ROWS - 1) / RG_ROWS), (unsigned)((a.nmb + RG_BLKS - 1) / RG_BLKS));
    hipLaunchKernelGGL(rows_gemm_kernel<true>, grid, dim3(512), 0, as_stream(stream), a);
    return check_launch("fp_mlp_fused_pre(gemm)");
}

}  // namespace pdm
