// Per-row MLP over many rows with fp32 operands EMULATED on the bf16 matrix pipe (gfx950) — opt-in, not the default path.
//
//   out[r] = W_3 relu(W_2 relu(W_1 in[r] + b_1) + b_2) + b_3        rows (R, cin) contiguous fp32 in, fp32 out
//
// Semantics: make_fc_layers of /root/reference/pcdet/models/dense_heads/point_head_template.py:35-48 in eval mode
// (BatchNorm folded on the host), the same chain rows_chain.hip runs on v_mfma_f32_16x16x4_f32.  That instruction is an
// exact fp32 FMA chain but runs at 1/16 of the bf16 rate of this chip, and the point head (209 GFLOP per step) is bound by
// it.  Here every fp32 operand is split into THREE bf16 pieces
//       v ~ hi + mid + lo,   hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid)       (8 + 8 + 8 significand bits;
//                                                              |v - (hi + mid + lo)| <= 2^-24 |v|: a residual may need a 9th bit)
// and a product a*b is formed from the six leading partial products
//       hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi          (dropped: mid*lo, lo*mid, lo*lo <= 2^-24 |a b|)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6 bf16 MFMAs of 16 cycles for 32 k against 8 fp32 MFMAs of 32 cycles
// = 3/8 of the matrix-pipe time.  The weights are split on the host (pack_layers_x3 in fused.py); activations are split in
// registers between the layers (3 v_cvt_pk_bf16_f32 + 4 subtractions + 4 bit operations per pair of values).
//
// Structure = rows_chain.hip's: a wave owns 16 rows and ALL channels of a layer; the D fragment of output blocks (2 q, 2 q + 1)
// of a layer — lane (pos, g) holds channels 16 mb + 4 g + i of row pos — is exactly the B fragment of k-block q of the next
// layer (k slot 8 g + j <-> channel 32 q + 16 (j >> 2) + 4 g + (j & 3); the host packs A to match), so activations stay in
// registers; the four waves of a workgroup share the weight stream through LDS: chunks of 24 fragments (4 output blocks x
// 2 k-blocks x 3 pieces, 24 KB), double-buffered, one barrier per chunk (48 MFMAs per wave).  The chunks of the three
// layers lie back to back in memory in the order they are consumed, so the stream is one pointer that wraps per tile.
//
// Measured (bs = 32 x 16384 rows, both stacks of the point head): 1.03-1.07 ms against 1.60 ms on the fp32 instruction (1.5x;
// 196-200 TFLOP/s fp32-equivalent), constant 1.0 ns per row from 65 k to 2 M rows.  The six-product arithmetic alone is 0.51 ms at
// the nominal bf16 rate; timing builds (X3_DIAG): weight fetch cache-hot -5 %, no barriers 0, no re-splitting -3 %, no fragment
// reads -6 %, all four together 0.72 ms — i.e. a bare MFMA stream in this structure reaches ~70 % of the nominal rate and the
// operand delivery around it costs the rest (profiles/r04k_split_bf16_chain_diag.txt).
#include "common.h"

namespace pdm {

typedef float x3_f4 __attribute__((ext_vector_type(4)));
typedef unsigned x3_u4 __attribute__((ext_vector_type(4)));
typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x3_f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) x3_u4 *x3_gu4c;

constexpr int X3_THREADS = 256;
#ifndef X3_PAIR
#define X3_PAIR 0         // 1: two output blocks per step, their MFMAs interleaved.  Measured: NOT faster — the bare MFMA stream
                          // (X3_DIAG 15) 0.755 against 0.707 ms, the product 1.14 ms with 18 spilled registers against 1.03-1.07:
                          // six dependent v_mfma_f32_16x16x32_bf16 in a row issue at the rate of independent ones
#endif
#define X3_AN (X3_PAIR ? 6 : 3)
#define an2 an
#ifndef X3_DIAG
#define X3_DIAG 0         // timing builds (results wrong): 1 every fetch reads chunk 0 (cache-hot), 2 no barriers, 4 no re-splitting of activations, 8 no fragment reads
#endif
#ifndef X3_AHEAD
#define X3_AHEAD 0        // 1: request the next tile's input rows under the current tile's second layer — measured: 256 VGPRs with 12
                          // spilled, 1.07-1.10 ms for the point head's two stacks against 1.03-1.07 without (tools/diag/x3_rate.py)
#endif
constexpr int X3_CHUNK_U4 = 24 * 64;        // 24 fragments x 64 lanes x 16 bytes

struct RowsChainX3Args {
    int rows, in_stride;
    const float *in;
    const unsigned *wstream;     // chunks of the three layers back to back (fused.py::pack_layers_x3)
    const float *bias;           // fp32, padded widths, layers back to back
    int boff[3];
    float *out;
    int out_stride, cout, relu_last;
};

__device__ __forceinline__ unsigned x3_pack2(float a, float b) {
    const x3_f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x3_bf16x2));   // v_cvt_pk_bf16_f32 (round to nearest even)
}
// two fp32 values -> their three bf16 pieces, packed pairwise (a in the low half)
__device__ __forceinline__ void x3_split2(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo) {
    hi = x3_pack2(a, b);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);      // exact
    mid = x3_pack2(ra, rb);
    const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);  // exact
    lo = x3_pack2(sa, sb);
}
// the B fragment of one k-block (32 channels): the lane's float4 of output block 2 q and of block 2 q + 1
__device__ __forceinline__ void x3_split8(const x3_f4 &u, const x3_f4 &v, x3_u4 (&pl)[3]) {
    unsigned h[4], m[4], l[4];
    x3_split2(u.x, u.y, h[0], m[0], l[0]);
    x3_split2(u.z, u.w, h[1], m[1], l[1]);
    x3_split2(v.x, v.y, h[2], m[2], l[2]);
    x3_split2(v.z, v.w, h[3], m[3], l[3]);
    pl[0] = x3_u4{h[0], h[1], h[2], h[3]};
    pl[1] = x3_u4{m[0], m[1], m[2], m[3]};
    pl[2] = x3_u4{l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ x3_f4 x3_mfma(const x3_u4 &a, const x3_u4 &b, const x3_f4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x3_bf16x8, a), __builtin_bit_cast(x3_bf16x8, b), c, 0, 0, 0);
}
// one output block x one k-block: the six leading partial products, small ones first
__device__ __forceinline__ void x3_mac(x3_f4 &acc, const x3_u4 (&a)[3], const x3_u4 (&b)[3]) {
    acc = x3_mfma(a[2], b[0], acc);   // lo  * hi
    acc = x3_mfma(a[0], b[2], acc);   // hi  * lo
    acc = x3_mfma(a[1], b[1], acc);   // mid * mid
    acc = x3_mfma(a[1], b[0], acc);   // mid * hi
    acc = x3_mfma(a[0], b[1], acc);   // hi  * mid
    acc = x3_mfma(a[0], b[0], acc);   // hi  * hi
}

// The weight stream: chunk `c` of the `total` chunks of the network sits at wstream + c * 24 KB.  fetch = this thread's six
// 16-byte pieces of a chunk into registers; stash = into an LDS buffer (same layout as in memory).
__device__ __forceinline__ void x3_fetch(x3_u4 (&r)[6], const unsigned *__restrict__ wstream, int c, int t) {
    x3_gu4c src = (x3_gu4c)(wstream) + (size_t)((X3_DIAG & 1) ? 0 : c) * X3_CHUNK_U4 + t;
#pragma unroll
    for (int u = 0; u < 6; ++u) r[u] = src[u * X3_THREADS];
}
__device__ __forceinline__ void x3_stash(const x3_u4 (&r)[6], x3_u4 *buf, int t) {
#pragma unroll
    for (int u = 0; u < 6; ++u) buf[u * X3_THREADS + t] = r[u];
}

// One chunk = 8 steps (a step = one output block x one k-block = 3 fragments, 6 MFMAs).  The A fragments of step s + 1 are read
// while step s's MFMAs run; the chunk's barrier sits in front of its LAST step: the next chunk is stashed into the other
// buffer, barrier, and step 0 of the next chunk is read from there under the last 6 MFMAs — no LDS latency is exposed at a
// chunk boundary (rows_chain.hip's protocol).  WAR: the other buffer held chunk c - 1, last read in step 6 of chunk c - 1, in
// front of that chunk's barrier.  `an` = the fragments of the step about to run (in: this chunk's step 0; out: the next chunk's).
#define X3_STEP(S, ACC, BFRAG)                                                                                     \
    {                                                                                                              \
        x3_u4 a_[3];                                                                                               \
        _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_) a_[pl_] = an[pl_];                                    \
        if ((S) < 7) {                                                                                             \
            if (!(X3_DIAG & 8)) { _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_) an[pl_] = buf[(((S) + 1) * 3 + pl_) * 64]; } \
        } else {                                                                                                   \
            x3_stash(r, lds + (p ^ 1) * X3_CHUNK_U4, t);                                                           \
            if (!(X3_DIAG & 2)) __syncthreads();                                                                   \
            _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_) an[pl_] = nbuf[pl_ * 64];                          \
        }                                                                                                          \
        x3_mac(ACC, a_, BFRAG);                                                                                    \
    }

// Two steps at once (X3_PAIR): the 12 MFMAs of two output blocks alternate between the two accumulators, so no MFMA waits for the
// one issued just before it; `an` holds the six fragments of the pair about to run.
__device__ __forceinline__ void x3_mac2(x3_f4 &c0, x3_f4 &c1, const x3_u4 (&a)[6], const x3_u4 (&b0)[3], const x3_u4 (&b1)[3]) {
    c0 = x3_mfma(a[2], b0[0], c0); c1 = x3_mfma(a[5], b1[0], c1);
    c0 = x3_mfma(a[0], b0[2], c0); c1 = x3_mfma(a[3], b1[2], c1);
    c0 = x3_mfma(a[1], b0[1], c0); c1 = x3_mfma(a[4], b1[1], c1);
    c0 = x3_mfma(a[1], b0[0], c0); c1 = x3_mfma(a[4], b1[0], c1);
    c0 = x3_mfma(a[0], b0[1], c0); c1 = x3_mfma(a[3], b1[1], c1);
    c0 = x3_mfma(a[0], b0[0], c0); c1 = x3_mfma(a[3], b1[0], c1);
}
#define X3_STEP2(S, ACC0, ACC1, B0, B1)                                                                            \
    {                                                                                                              \
        x3_u4 a_[6];                                                                                               \
        _Pragma("unroll") for (int f_ = 0; f_ < 6; ++f_) a_[f_] = an2[f_];                                        \
        if ((S) < 6) {                                                                                             \
            if (!(X3_DIAG & 8)) { _Pragma("unroll") for (int f_ = 0; f_ < 6; ++f_) an2[f_] = buf[(((S) + 2) * 3 + f_) * 64]; } \
        } else {                                                                                                   \
            x3_stash(r, lds + (p ^ 1) * X3_CHUNK_U4, t);                                                           \
            if (!(X3_DIAG & 2)) __syncthreads();                                                                   \
            _Pragma("unroll") for (int f_ = 0; f_ < 6; ++f_) an2[f_] = nbuf[f_ * 64];                              \
        }                                                                                                          \
        x3_mac2(ACC0, ACC1, a_, B0, B1);                                                                           \
    }

// One layer with NK k-blocks (32 channels each) in and NMB >= 4 output blocks (16 channels each): NMB / 4 x NK / 2 chunks, k inner;
// chunk layout: step (kbi, i) = kbi * 4 + i.  On entry buffer p holds this layer's first chunk and `an` its step 0; on exit the
// same for what follows in the stream.  `c` = running chunk number (wraps at `total`).
template <int NK, int NMB>
__device__ __forceinline__ void x3_layer(const x3_u4 (&in)[NK][3], x3_f4 (&acc)[NMB], const float *__restrict__ bias,
                                         const unsigned *__restrict__ wstream, int &c, int total, x3_u4 *lds, int &p, int t, int lane,
                                         x3_u4 (&an)[X3_AN]) {
    static_assert(NK % 2 == 0 && NMB % 4 == 0, "x3_layer: whole chunks");
    const int g = lane >> 4;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) acc[mb] = *reinterpret_cast<const x3_f4 *>(bias + 16 * mb + 4 * g);
#pragma unroll
    for (int mg = 0; mg < NMB / 4; ++mg) {
#pragma unroll
        for (int kg = 0; kg < NK / 2; ++kg) {
            x3_u4 r[6];
            const int nxt = c + 1 == total ? 0 : c + 1;
            x3_fetch(r, wstream, nxt, t);
            const x3_u4 *buf = lds + p * X3_CHUNK_U4 + lane, *nbuf = lds + (p ^ 1) * X3_CHUNK_U4 + lane;
#if X3_PAIR
#pragma unroll
            for (int kbi = 0; kbi < 2; ++kbi) {
#pragma unroll
                for (int i = 0; i < 4; i += 2)
                    X3_STEP2(kbi * 4 + i, acc[4 * mg + i], acc[4 * mg + i + 1], in[2 * kg + kbi], in[2 * kg + kbi])
            }
#else
#pragma unroll
            for (int kbi = 0; kbi < 2; ++kbi) {
#pragma unroll
                for (int i = 0; i < 4; ++i) X3_STEP(kbi * 4 + i, acc[4 * mg + i], in[2 * kg + kbi])
            }
#endif
            p ^= 1;
            c = nxt;
        }
    }
}
// The heads' last layer: ONE output block over 8 k-blocks = one chunk, step = k-block.
template <int NK>
__device__ __forceinline__ void x3_last_layer(const x3_u4 (&in)[NK][3], x3_f4 &acc, const float *__restrict__ bias,
                                              const unsigned *__restrict__ wstream, int &c, int total, x3_u4 *lds, int &p, int t, int lane,
                                              x3_u4 (&an)[X3_AN]) {
    static_assert(NK == 8, "x3_last_layer: 8 k-blocks x 3 pieces = one chunk");
    const int g = lane >> 4;
    acc = *reinterpret_cast<const x3_f4 *>(bias + 4 * g);
    x3_u4 r[6];
    const int nxt = c + 1 == total ? 0 : c + 1;
    x3_fetch(r, wstream, nxt, t);
    const x3_u4 *buf = lds + p * X3_CHUNK_U4 + lane, *nbuf = lds + (p ^ 1) * X3_CHUNK_U4 + lane;
#if X3_PAIR
    x3_f4 acc1 = x3_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NK; kb += 2) X3_STEP2(kb, acc, acc1, in[kb], in[kb + 1])
    acc += acc1;
#else
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) X3_STEP(kb, acc, in[kb])
#endif
    p ^= 1;
    c = nxt;
}
// bias is in the accumulator already: ReLU, then the next layer's B fragments
template <int NMB>
__device__ __forceinline__ void x3_relu_split(const x3_f4 (&acc)[NMB], x3_u4 (&out)[NMB / 2][3]) {
#pragma unroll
    for (int q = 0; q < NMB / 2; ++q) {
        x3_f4 u = acc[2 * q], v = acc[2 * q + 1];
        u.x = fmaxf(u.x, 0.f); u.y = fmaxf(u.y, 0.f); u.z = fmaxf(u.z, 0.f); u.w = fmaxf(u.w, 0.f);
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (X3_DIAG & 4) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) out[q][pl] = x3_u4{__float_as_uint(u.x), __float_as_uint(u.y), __float_as_uint(v.x), __float_as_uint(v.y)};
        } else
        x3_split8(u, v, out[q]);
    }
}

// NK0 = k-blocks (32 channels) of the input, NM1 / NM2 = output blocks (16 channels) of layers 1 / 2; layer 3 has one block.
template <int NK0, int NM1, int NM2>
__global__ __launch_bounds__(X3_THREADS, 2) void rows_chain_x3_kernel(RowsChainX3Args a) {
    static_assert(NM2 == 16, "the last layer's chunk holds 8 k-blocks");
    __shared__ __attribute__((aligned(16))) x3_u4 lds[2 * X3_CHUNK_U4];   // 48 KB
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int pos = lane & 15, g = lane >> 4;
    constexpr int TOTAL = (NM1 / 4) * (NK0 / 2) + (NM2 / 4) * (NM1 / 4) + 1;
    const unsigned *wstream = a.wstream;
    const long long ntiles = ((long long)a.rows + 63) / 64;
    int p = 0, c = 0;
    {
        x3_u4 r[6];
        x3_fetch(r, wstream, 0, t);
        x3_stash(r, lds, t);
    }
    __syncthreads();
    x3_u4 an[X3_AN];
#pragma unroll
    for (int pl = 0; pl < X3_AN; ++pl) an[pl] = lds[pl * 64 + lane];
    x3_f4 xn[2 * NK0];       // the next tile's input rows (raw fp32), requested under this tile's second layer
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        asm volatile("" : "+s"(wstream));     // keeps the chunk addresses from being hoisted out of the loop as ~50 invariants
        long long row = tl * 64 + 16 * wave + pos;
        const bool live = row < a.rows;
        if (!live) row = a.rows - 1;
        x3_u4 x0[NK0][3];
        if (X3_AHEAD && tl != (long long)blockIdx.x) {
#pragma unroll
            for (int kb = 0; kb < NK0; ++kb) x3_split8(xn[2 * kb], xn[2 * kb + 1], x0[kb]);
        } else {
            const float *__restrict__ src = a.in + (size_t)row * a.in_stride + 4 * g;
#pragma unroll
            for (int kb = 0; kb < NK0; ++kb) {
                const x3_f4 u = *reinterpret_cast<const x3_f4 *>(src + 32 * kb), v = *reinterpret_cast<const x3_f4 *>(src + 32 * kb + 16);
                x3_split8(u, v, x0[kb]);
            }
        }
        x3_u4 x1[NM1 / 2][3];
        {
            x3_f4 acc[NM1];
            x3_layer<NK0, NM1>(x0, acc, a.bias + a.boff[0], wstream, c, TOTAL, lds, p, t, lane, an);
            x3_relu_split<NM1>(acc, x1);
        }
        if (X3_AHEAD) {
            long long nrow = (tl + gridDim.x) * 64 + 16 * wave + pos;
            if (nrow >= a.rows) nrow = a.rows - 1;
            const float *__restrict__ nsrc = a.in + (size_t)nrow * a.in_stride + 4 * g;
#pragma unroll
            for (int kb = 0; kb < NK0; ++kb) {
                xn[2 * kb] = *reinterpret_cast<const x3_f4 *>(nsrc + 32 * kb);
                xn[2 * kb + 1] = *reinterpret_cast<const x3_f4 *>(nsrc + 32 * kb + 16);
            }
        }
        x3_u4 x2[NM2 / 2][3];
        {
            x3_f4 acc[NM2];
            x3_layer<NM1 / 2, NM2>(x1, acc, a.bias + a.boff[1], wstream, c, TOTAL, lds, p, t, lane, an);
            x3_relu_split<NM2>(acc, x2);
        }
        x3_f4 y;
        x3_last_layer<NM2 / 2>(x2, y, a.bias + a.boff[2], wstream, c, TOTAL, lds, p, t, lane, an);
        if (a.relu_last) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        if (live) {
            float *__restrict__ orow = a.out + (size_t)row * a.out_stride;
            const int c0 = 4 * g;
            if (c0 + 4 <= a.cout) *reinterpret_cast<x3_f4 *>(orow + c0) = y;
            else {
                if (c0 < a.cout) orow[c0] = y.x;
                if (c0 + 1 < a.cout) orow[c0 + 1] = y.y;
                if (c0 + 2 < a.cout) orow[c0 + 2] = y.z;
            }
        }
    }
}

static int g_x3_wg_per_cu = 12;
}  // namespace pdm

using namespace pdm;

extern "C" int pdm_tune_rows_x3_wg_per_cu(int n) { const int old = g_x3_wg_per_cu; if (n > 0) g_x3_wg_per_cu = n; return old; }

// Bytes of the weight stream pdm_rows_mlp_x3 expects for `dims` (nlayers + 1 padded widths), 0 when no instantiation fits.
extern "C" size_t pdm_rows_mlp_x3_stream_bytes(int nlayers, const int *dims) {
    if (nlayers != 3 || !dims || dims[0] != 128 || dims[1] != 256 || dims[2] != 256 || dims[3] != 16) return 0;
    return (size_t)((256 / 64) * (128 / 64) + (256 / 64) * (256 / 64) + 1) * X3_CHUNK_U4 * 16;
}

// Per-row three-layer MLP with split-bf16 ("3 x bf16, 6 products, fp32 accumulation") emulation of fp32 — see the file header.
// dims = {128, 256, 256, 16} only (the point head's stacks); wstream from fused.py::pack_layers_x3; bias fp32 padded, layers back
// to back.  Error against exact fp32 arithmetic: <= ~3 x 2^-24 per product relative to |a b| (tests/test_x3_gpu.py states the bound).
extern "C" int pdm_rows_mlp_x3(void *stream, int rows, int cin, const float *in_pm, int nlayers, const int *dims,
                               const void *wstream, size_t wstream_bytes, const float *bias, int relu_last, float *out_pm,
                               int out_stride, int cout) {
    PDM_REQUIRE(rows >= 0 && cin >= 1, PDM_E_BADARG, "rows_mlp_x3: rows=%d cin=%d", rows, cin);
    if (rows == 0) return 0;
    PDM_REQUIRE(in_pm && dims && wstream && bias && out_pm, PDM_E_BADARG, "rows_mlp_x3: null pointer");
    const size_t need = pdm_rows_mlp_x3_stream_bytes(nlayers, dims);
    PDM_REQUIRE(need != 0 && cin == dims[0], PDM_E_BADARG, "rows_mlp_x3: only 128 -> 256 -> 256 -> <= 16 is instantiated");
    PDM_REQUIRE(wstream_bytes >= need, PDM_E_BADARG, "rows_mlp_x3: weight stream of %zu bytes, need %zu", wstream_bytes, need);
    PDM_REQUIRE(cout > 0 && cout <= dims[3] && cout <= out_stride && out_stride % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(in_pm) | reinterpret_cast<uintptr_t>(wstream) | reinterpret_cast<uintptr_t>(bias) |
                      reinterpret_cast<uintptr_t>(out_pm)) & 15) == 0,
                PDM_E_BADARG, "rows_mlp_x3: buffers must be 16-byte aligned, out_stride a multiple of 4, cout <= 16");
    RowsChainX3Args a{};
    a.rows = rows; a.in_stride = cin; a.in = in_pm; a.wstream = static_cast<const unsigned *>(wstream); a.bias = bias;
    a.boff[0] = 0; a.boff[1] = dims[1]; a.boff[2] = dims[1] + dims[2];
    a.out = out_pm; a.out_stride = out_stride; a.cout = cout; a.relu_last = relu_last;
    const long long tiles = ((long long)rows + 63) / 64;
    const int grid = (int)(tiles < 256 * g_x3_wg_per_cu ? tiles : 256 * g_x3_wg_per_cu);
    hipLaunchKernelGGL((rows_chain_x3_kernel<4, 16, 16>), dim3(grid), dim3(X3_THREADS), 0, as_stream(stream), a);
    return check_launch("rows_mlp_x3");
}
