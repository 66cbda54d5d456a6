// Per-row MLP over MANY rows, whole chain in registers (fp32 MFMA, gfx950).
//
//   out[r] = act(W_L ... relu(W_1 in[r] + b_1) ... + b_L)        rows (R, cin) contiguous, R ~ 10^5 .. 10^6
//
// Used by pdm_rows_mlp_fused for the hybrid head: the point head's two 128 -> 256 -> 256 -> {3, 8} MLPs over every
// point (B * N = 524288 rows at the bench shape, 206 GFLOP per step: the largest contraction of the forward) and the
// heat-map head's per-cell 1x1 stack.  Semantics: make_fc_layers of
// /root/reference/pcdet/models/dense_heads/point_head_template.py:35-48 in eval mode (BatchNorm folded on the host).
//
// Why another kernel: the chain kernels of fused_mlp.hip give a workgroup 16-32 rows and stream every weight fragment
// from L2 per 4-8 MFMAs (measured 72 TFLOP/s on this shape, waves waiting on operand delivery).  Here
//   * a wave owns ONE tile of 16 rows and ALL output channels of a layer: the D fragment of output block mb (lane
//     (pos, g) holds channels 16 mb + 4 g + i of row pos) is exactly the B fragment the next layer reads for k-block
//     mb, so activations never leave the registers between layers (in[<=16] + acc[<=16] float4 per lane);
//   * the four waves of a workgroup (64 rows) share every weight fragment through LDS: the packed weights stream
//     through two 16 KB buffers in chunks of (4 output blocks x 4 k-blocks) = 16 fragments of 1 KB, loaded to
//     registers one chunk ahead and written to the other buffer behind the MFMAs: one barrier per 64 MFMAs per wave;
//   * everything is unrolled (the register arrays need constant indices): one instantiation per chain of widths.
// Bound: fp32 MFMA pipe (16 MFMAs per 4 ds_read_b128; LDS 32 B/clk/CU, L2 -> LDS 16 KB per 2048 pipe cycles).
#include "common.h"

namespace pdm {

typedef float f4 __attribute__((ext_vector_type(4)));

struct RowsChainArgs {
    int rows, in_stride;        // floats between input rows (= cin, a multiple of 16)
    const float *in;
    const float *wpack, *bias;  // packed as fused.py::pack_layer, layers back to back
    int woff[3], boff[3];       // float offsets of each layer
    float *out;
    int out_stride, cout, relu_last;
    // depthwise prologue (DW instantiations): a row is a cell of a channels-last (B, H, W, C) map and the chain's input
    // is relu(depthwise3x3(map)[cell] + shift) formed on the fly (bev_head.hip's kernel, same fma order)
    int dw_H, dw_W;
    const float *dw_w, *dw_shift;   // (9, C) tap-major with the BatchNorm scale folded in, (C)
};

constexpr int RC_THREADS = 256;
constexpr int RC_CHUNK_F4 = 16 * 64;   // 16 fragments x 64 lanes

// chunk (mb0 .. mb0 + nmb - 1) x (kb0 .. kb0 + 3) of a layer with NKB k-blocks: thread t fetches lane t % 64 of
// fragments (mb0 + i, kb0 + t / 64)
__device__ __forceinline__ void rc_fetch(f4 (&r)[4], const f4 *__restrict__ w, int nkb, int mb0, int kb0, int nmb, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nmb) r[i] = (w + ((size_t)(mb0 + i) * nkb + kb0) * 64)[t];   // uniform base (SGPRs) + one lane offset
}
__device__ __forceinline__ void rc_stash(const f4 (&r)[4], f4 *buf, int nmb, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nmb) buf[i * 256 + t] = r[i];
}

// One layer: in[NKB] -> acc[NMB] (bias added, floored).  `p` = which LDS buffer holds this layer's first chunk.
// next_*: the first chunk of the following layer (prefetched behind this layer's last chunk), next_w == nullptr: none.
template <int NKB, int NMB>
__device__ __forceinline__ void rc_layer(const f4 (&in)[NKB], f4 (&acc)[NMB], const f4 *__restrict__ w, const float *__restrict__ bias,
                                         f4 *lds, int &p, float floor, const f4 *__restrict__ next_w, int next_nkb, int next_nmb,
                                         int t, int lane, f4 (&r)[4]) {
    static_assert(NKB % 4 == 0, "rows_chain: layer inputs are multiples of 64 channels");
    constexpr int KG = NKB / 4, MG = (NMB + 3) / 4, NCH = KG * MG;
    const int g = lane >> 4;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) acc[mb] = *reinterpret_cast<const f4 *>(bias + 16 * mb + 4 * g);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int mb0 = (c / KG) * 4, kb0 = (c % KG) * 4;
        constexpr int dummy = 0; (void)dummy;
        const int nmb = NMB - mb0 < 4 ? NMB - mb0 : 4;
        // next chunk -> registers
        int fn = 0;
        if (c + 1 < NCH) {
            const int nm0 = ((c + 1) / KG) * 4, nk0 = ((c + 1) % KG) * 4;
            fn = NMB - nm0 < 4 ? NMB - nm0 : 4;
            rc_fetch(r, w, NKB, nm0, nk0, fn, t);
        } else if (next_w) {
            fn = next_nmb < 4 ? next_nmb : 4;
            rc_fetch(r, next_w, next_nkb, 0, 0, fn, t);
        }
        // this chunk: 4 k-blocks x nmb output blocks
        const f4 *buf = lds + p * RC_CHUNK_F4 + lane;
#pragma unroll
        for (int kbi = 0; kbi < 4; ++kbi) {
            f4 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmb) a[i] = buf[(i * 4 + kbi) * 64];
            const f4 b = in[kb0 + kbi];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmb) acc[mb0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b.x, acc[mb0 + i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmb) acc[mb0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b.y, acc[mb0 + i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmb) acc[mb0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b.z, acc[mb0 + i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmb) acc[mb0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b.w, acc[mb0 + i], 0, 0, 0);
            // keeps the fragment loads of later k-blocks from piling up in registers (an explicit one-ahead
            // prefetch of the next k-block's fragments measured no faster: the second wave of the SIMD covers the gap)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (fn) rc_stash(r, lds + (p ^ 1) * RC_CHUNK_F4, fn, t);
        __syncthreads();
        p ^= 1;
    }
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        acc[mb].x = __builtin_amdgcn_fmed3f(acc[mb].x, floor, __builtin_inff());
        acc[mb].y = __builtin_amdgcn_fmed3f(acc[mb].y, floor, __builtin_inff());
        acc[mb].z = __builtin_amdgcn_fmed3f(acc[mb].z, floor, __builtin_inff());
        acc[mb].w = __builtin_amdgcn_fmed3f(acc[mb].w, floor, __builtin_inff());
    }
}

// NK0 = k-blocks (16 channels) of the input, NK1 .. NK3 = output blocks of layers 1 .. 3 (0 = layer absent)
template <int NK0, int NK1, int NK2, int NK3, bool DW = false>
__global__ __launch_bounds__(RC_THREADS, 2) void rows_chain_kernel(RowsChainArgs a) {
    __shared__ __attribute__((aligned(16))) f4 lds[2 * RC_CHUNK_F4];
    __shared__ __attribute__((aligned(16))) f4 dww[DW ? 10 * NK0 * 4 : 1];   // 9 taps + shift, C / 4 quads each
    if constexpr (DW) {
        for (int i = threadIdx.x; i < 9 * NK0 * 4; i += RC_THREADS) dww[i] = reinterpret_cast<const f4 *>(a.dw_w)[i];
        for (int i = threadIdx.x; i < NK0 * 4; i += RC_THREADS) dww[9 * NK0 * 4 + i] = reinterpret_cast<const f4 *>(a.dw_shift)[i];
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int pos = lane & 15, g = lane >> 4;
    const f4 *w1 = reinterpret_cast<const f4 *>(a.wpack + a.woff[0]);
    const f4 *w2 = NK2 ? reinterpret_cast<const f4 *>(a.wpack + a.woff[1]) : nullptr;
    const f4 *w3 = NK3 ? reinterpret_cast<const f4 *>(a.wpack + a.woff[2]) : nullptr;
    constexpr int NL = NK3 ? 3 : NK2 ? 2 : 1;
    const float neg_inf = -__builtin_inff();
    f4 r[4];
    __shared__ __attribute__((aligned(16))) f4 halo[DW ? 6 * 18 * 4 : 1];   // DW: (4 + 2) x (16 + 2) cells x 16 channels
    __shared__ __attribute__((aligned(16))) f4 xs[DW ? 64 * (NK0 * 4 + 1) : 1];   // DW: the patch's convolved rows (+1 quad pad)
    const int ntx = DW ? (a.dw_W + 15) / 16 : 1, nty = DW ? (a.dw_H + 3) / 4 : 1;
    const long long ntiles = DW ? (long long)(a.rows / (a.dw_H * a.dw_W)) * nty * ntx : ((long long)a.rows + 63) / 64;
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long long tile0 = tl * 64;
        // (the weight pointers pass through an empty asm so the ~140 chunk addresses are formed inside the loop with
        //  scalar adds instead of being hoisted out of it as loop invariants, where they would take every register)
        asm volatile("" : "+s"(w1), "+s"(w2), "+s"(w3));
        // first chunk of layer 1 (the barrier of the previous iteration's last chunk makes the buffer free)
        int p = 0;
        rc_fetch(r, w1, NK0, 0, 0, NK1 < 4 ? NK1 : 4, t);
        rc_stash(r, lds, NK1 < 4 ? NK1 : 4, t);
        // this lane's input row: channels [16 kb + 4 g, +4) of row tile0 + 16 wave + pos
        // (DW: the workgroup owns a 4 x 16 patch of cells of one image, wave = patch row, pos = cell in it)
        long long row = tile0 + 16 * wave + pos;
        bool live = row < a.rows;
        int dw_b = 0, dw_y0 = 0, dw_x0 = 0;
        if constexpr (DW) {
            const int tx = (int)(tl % ntx), ty = (int)((tl / ntx) % nty);
            dw_b = (int)(tl / ((long long)ntx * nty));
            dw_y0 = ty * 4; dw_x0 = tx * 16;
            live = dw_y0 + wave < a.dw_H && dw_x0 + pos < a.dw_W;
            row = ((long long)dw_b * a.dw_H + dw_y0 + wave) * a.dw_W + dw_x0 + pos;
        }
        if (!live) row = a.rows - 1;
        const long long out_row = row;
        f4 x0[NK0];
        if constexpr (!DW) {
            const float *__restrict__ src = a.in + (size_t)row * a.in_stride + 4 * g;
#pragma unroll
            for (int kb = 0; kb < NK0; ++kb) x0[kb] = *reinterpret_cast<const f4 *>(src + 16 * kb);
            __syncthreads();
        } else {
            // depthwise 3x3 + shift + ReLU: the patch's halo is staged through LDS 16 channels at a time (each cell of
            // the map is fetched once per workgroup instead of up to nine times), the next slice's loads in flight
            // under the current slice's taps
            constexpr int C4 = NK0 * 4;
            const f4 *__restrict__ img = reinterpret_cast<const f4 *>(a.in) + (size_t)dw_b * a.dw_H * a.dw_W * C4;
            f4 hv[2];
            auto fetch = [&](int kb) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = t + u * RC_THREADS;           // f4 index in the halo slice: (cell, quad)
                    hv[u] = f4{0.f, 0.f, 0.f, 0.f};
                    if (i < 6 * 18 * 4) {
                        const int cell = i >> 2, quad = i & 3;
                        const int gy = dw_y0 - 1 + cell / 18, gx = dw_x0 - 1 + cell % 18;
                        if (gy >= 0 && gy < a.dw_H && gx >= 0 && gx < a.dw_W) hv[u] = img[((size_t)gy * a.dw_W + gx) * C4 + kb * 4 + quad];
                    }
                }
            };
            fetch(0);
            // a runtime loop over the 16-channel slices (unrolled, the compiler gathers all 80 tap reads up front and
            // spills them); a lane parks its slice results in its own LDS row and reads them back as x0[] afterwards
            f4 *mine = xs + (wave * 16 + pos) * (C4 + 1) + g;
#pragma unroll 1
            for (int kb = 0; kb < NK0; ++kb) {
                if (t < 6 * 18 * 4) halo[t] = hv[0];
                if (t + RC_THREADS < 6 * 18 * 4) halo[t + RC_THREADS] = hv[1];
                __syncthreads();
                if (kb + 1 < NK0) fetch(kb + 1);
                f4 acc = dww[9 * C4 + kb * 4 + g];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const f4 v = halo[((wave + dy) * 18 + pos + dx) * 4 + g];
                        const f4 k = dww[(dy * 3 + dx) * C4 + kb * 4 + g];
                        acc.x = fmaf(k.x, v.x, acc.x); acc.y = fmaf(k.y, v.y, acc.y);
                        acc.z = fmaf(k.z, v.z, acc.z); acc.w = fmaf(k.w, v.w, acc.w);
                    }
                mine[kb * 4] = f4{fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f)};
                __syncthreads();
            }
#pragma unroll
            for (int kb = 0; kb < NK0; ++kb) x0[kb] = mine[kb * 4];
        }
        float *__restrict__ orow = live ? a.out + (size_t)out_row * a.out_stride : nullptr;
        auto store = [&](auto &y, int nmb) {
            if (!orow) return;
#pragma unroll
            for (int mb = 0; mb < nmb; ++mb) {
                const int c0 = 16 * mb + 4 * g;
                if (c0 + 4 <= a.cout) *reinterpret_cast<f4 *>(orow + c0) = y[mb];
                else {
                    if (c0 < a.cout) orow[c0] = y[mb].x;
                    if (c0 + 1 < a.cout) orow[c0 + 1] = y[mb].y;
                    if (c0 + 2 < a.cout) orow[c0 + 2] = y[mb].z;
                }
            }
        };
        f4 x1[NK1];
        rc_layer<NK0, NK1>(x0, x1, w1, a.bias + a.boff[0], lds, p, (NL > 1 || a.relu_last) ? 0.0f : neg_inf, w2, NK1, NK2, t, lane, r);
        if constexpr (NL == 1) {
            store(x1, NK1);
        } else {
            f4 x2[NK2 ? NK2 : 1];
            rc_layer<NK1, (NK2 ? NK2 : 1)>(x1, x2, w2, a.bias + a.boff[1], lds, p, (NL > 2 || a.relu_last) ? 0.0f : neg_inf, w3, NK2, NK3, t, lane, r);
            if constexpr (NL == 2) {
                store(x2, NK2);
            } else {
                f4 x3[NK3 ? NK3 : 1];
                rc_layer<(NK2 ? NK2 : 4), (NK3 ? NK3 : 1)>(x2, x3, w3, a.bias + a.boff[2], lds, p, a.relu_last ? 0.0f : neg_inf, nullptr, 0, 0, t,
                                                          lane, r);
                store(x3, NK3);
            }
        }
    }
}

static bool rc_shape_is(int nlayers, const int *dims, int k0, int k1, int k2, int k3) {
    const int want[4] = {k0, k1, k2, k3};
    const int nl = k3 ? 3 : k2 ? 2 : 1;
    if (nlayers != nl) return false;
    for (int i = 0; i <= nl; ++i)
        if (dims[i] != 16 * want[i]) return false;
    return true;
}

// Returns 1 when the chain was launched, 0 when no instantiation fits (the caller takes the general kernel), < 0 / > 0
// HIP codes on error.  Preconditions checked by the caller: pointers non-null and 16-byte aligned, out_stride % 4 == 0.
int rows_chain_launch(void *stream, int rows, int cin, const float *in_pm, int nlayers, const int *dims, const float *wpack,
                      const float *bias, int relu_last, float *out_pm, int out_stride, int cout, int *launched) {
    *launched = 0;
    if (rows < 8192 || cin % 16 != 0 || nlayers < 1 || nlayers > 3 || dims[0] != cin) return 0;
    RowsChainArgs a{};
    a.rows = rows; a.in_stride = cin; a.in = in_pm; a.wpack = wpack; a.bias = bias;
    int wo = 0, bo = 0;
    for (int l = 0; l < nlayers; ++l) {
        a.woff[l] = wo; a.boff[l] = bo;
        wo += dims[l] * dims[l + 1];
        bo += dims[l + 1];
    }
    a.out = out_pm; a.out_stride = out_stride; a.cout = cout; a.relu_last = relu_last;
    const long long tiles = ((long long)rows + 63) / 64;
    const int grid = (int)(tiles < 256 * 12 ? tiles : 256 * 12);
#define RC_TRY(K0, K1, K2, K3)                                                                                           \
    if (rc_shape_is(nlayers, dims, K0, K1, K2, K3)) {                                                                    \
        hipLaunchKernelGGL((rows_chain_kernel<K0, K1, K2, K3>), dim3(grid), dim3(RC_THREADS), 0, as_stream(stream), a);  \
        *launched = 1;                                                                                                   \
        return check_launch("rows_mlp_fused(chain)");                                                                    \
    }
    RC_TRY(8, 16, 16, 1)    // point head: 128 -> 256 -> 256 -> <= 16 (class logits, box code)
    RC_TRY(8, 4, 4, 1)      // heat-map head per-cell stack: 128 -> 64 -> 64 -> <= 16
#undef RC_TRY
    return 0;
}

// The heat-map head's whole stack in one launch: depthwise 3x3 (+ folded BN + ReLU) as the chain's prologue.
// Returns 1 in *launched when an instantiation fits.
int rows_chain_dw_launch(void *stream, int B, int H, int W, int C, const float *map, const float *dw_w, const float *dw_shift,
                         int nlayers, const int *dims, const float *wpack, const float *bias, int relu_last, float *out_pm,
                         int out_stride, int cout, int *launched) {
    *launched = 0;
    const long long rows = (long long)B * H * W;
    if (rows < 8192 || rows >= (1ll << 31) || C % 16 != 0 || nlayers != 3 || dims[0] != C) return 0;
    RowsChainArgs a{};
    a.rows = (int)rows; a.in_stride = C; a.in = map; a.wpack = wpack; a.bias = bias;
    int wo = 0, bo = 0;
    for (int l = 0; l < nlayers; ++l) {
        a.woff[l] = wo; a.boff[l] = bo;
        wo += dims[l] * dims[l + 1];
        bo += dims[l + 1];
    }
    a.out = out_pm; a.out_stride = out_stride; a.cout = cout; a.relu_last = relu_last;
    a.dw_H = H; a.dw_W = W; a.dw_w = dw_w; a.dw_shift = dw_shift;
    const long long tiles = (long long)B * ((H + 3) / 4) * ((W + 15) / 16);
    const int grid = (int)(tiles < 256 * 12 ? tiles : 256 * 12);
    if (rc_shape_is(nlayers, dims, 8, 4, 4, 1)) {
        hipLaunchKernelGGL((rows_chain_kernel<8, 4, 4, 1, true>), dim3(grid), dim3(RC_THREADS), 0, as_stream(stream), a);
        *launched = 1;
        return check_launch("bev_head_fused");
    }
    return 0;
}

}  // namespace pdm

// Depthwise 3x3 + folded BN + ReLU over a channels-last (B, H, W, C) map followed by a per-cell MLP, in ONE kernel
// (the heat-map head in eval mode): out[cell] = MLP(relu(dw3x3(map)[cell] + shift)).  Only the shapes rows_chain.hip
// instantiates (C = 128 -> 64 -> 64 -> <= 16); PDM_E_BADARG otherwise (the caller then runs the two kernels apart).
extern "C" int pdm_bev_head_fused(void *stream, int B, int H, int W, int C, const float *map, const float *dw_w,
                                  const float *dw_shift, int nlayers, const int *dims, const float *wpack, const float *bias,
                                  int relu_last, float *out_pm, int out_stride, int cout) {
    using namespace pdm;
    PDM_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0, PDM_E_BADARG, "bev_head_fused: bad size");
    if (B == 0) return 0;
    PDM_REQUIRE(map && dw_w && dw_shift && dims && wpack && bias && out_pm, PDM_E_BADARG, "bev_head_fused: null pointer");
    PDM_REQUIRE(cout > 0 && cout <= out_stride && out_stride % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(map) | reinterpret_cast<uintptr_t>(dw_w) | reinterpret_cast<uintptr_t>(dw_shift) |
                      reinterpret_cast<uintptr_t>(wpack) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(out_pm)) & 15) == 0,
                PDM_E_BADARG, "bev_head_fused: buffers must be 16-byte aligned, out_stride a multiple of 4");
    int launched = 0;
    const int rc = rows_chain_dw_launch(stream, B, H, W, C, map, dw_w, dw_shift, nlayers, dims, wpack, bias, relu_last, out_pm,
                                        out_stride, cout, &launched);
    if (rc) return rc;
    PDM_REQUIRE(launched, PDM_E_BADARG, "bev_head_fused: no instantiation for these widths");
    return 0;
}
