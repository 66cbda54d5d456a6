// bf16 contractions of the TRAINING path on the matrix cores (gfx950): the shared MLPs' 1x1 convolutions / Linear
// layers, forward, data gradient and weight gradient, over channels-last rows.
//
// What they replace: in the reference every shared-MLP stage is torch's Conv2d(1x1) / Linear
// (/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:91-97, models/dense_heads/
// point_head_template.py:35-48), i.e. vendor GEMM / implicit-GEMM kernels.  Round 2 of this repo still called the
// vendor GEMMs (rocBLAS / hipBLASLt "Cijk_*") for these layers in training.
//
// Shapes: rows R = B * M * nsample (0.5 - 4 M), channels K, N in 16 .. 1536.  Every product is TALL AND SKINNY and
// therefore bound by streaming the activations once (2 (K + N) bytes per row against 2 K N flop: 32 - 128 flop per byte,
// far under the bf16 MFMA ridge), so the kernels are built around full-line coalesced 16-byte traffic and LDS-staged
// fragments, not around MFMA occupancy:
//   tg_nt_kernel     Y[R,N]  = X[R,K] . W[N,K]^T   (forward; data gradient with the transposed weights)
//                    128 x 128 (wide layers) or 256 x 64 / 256 x 32 (narrow ones) output tile per 256-thread workgroup, K in
//                    steps of 64 through ONE LDS stage with the next step's 16-byte loads already in registers (3 - 4
//                    workgroups per CU overlap each other's latency);
//                    A / B fragments by ds_read_b128 from XOR-swizzled 128-byte rows (conflict-free);
//                    v_mfma_f32_32x32x16_bf16, fp32 accumulation; the epilogue rounds to bf16 (RNE), transposes through
//                    LDS and stores whole 256-byte row pieces; optionally it also leaves per-column sums of y and y^2 of
//                    the ROUNDED outputs (the BatchNorm statistics of the next operator: one pass over Y saved).
//   tg_tn_kernel     dW[N,K] = dY[R,N]^T . X[R,K]   (weight gradient): the contraction runs over the ROWS, so both
//                    operands are needed "column-major"; the tiles are staged row-major as they lie in memory and the
//                    fragments come from ds_read_b64_tr_b16 (the hardware transposing LDS read); a workgroup owns a
//                    128 x 128 tile of dW for one slab of rows, slabs are summed in a fixed order (bit-reproducible).
// Numerics = the contract the bf16-emulating checker states (oracle/cpu_detector.py): operands are bf16, products are
// exact, accumulation is fp32, the forward / data-gradient result is rounded once to bf16, the weight gradient stays fp32.
#include "common.h"

namespace pdm {

typedef __bf16 tg_bf16x8 __attribute__((ext_vector_type(8)));
typedef short tg_s16x4 __attribute__((ext_vector_type(4)));
typedef short tg_s16x8 __attribute__((ext_vector_type(8)));
typedef float tg_f32x16 __attribute__((ext_vector_type(16)));

constexpr int TG_T = 256;
constexpr int TG_BK = 64;
constexpr int TG_WR = 64;            // rows per stage of the weight-gradient kernel
constexpr int TG_WPITCH = 320;       // bytes per LDS row there: 256 of data + 64 of pad (transposed reads conflict-free)

__device__ __forceinline__ unsigned short tg_bf16(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float tg_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// two fp32 -> two bf16 (round to nearest even) in one instruction, v_cvt_pk_bf16_f32: `a` in the low half.  (The software
// form costs ~8 VALU instructions per value; the forward kernel's epilogue was issue-bound on it: SQ_ACTIVE_INST_ANY 52 %.)
typedef __bf16 tg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned tg_pack2(float a, float b) {
    const tg_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, tg_bf16x2));
}

// Workgroups are dealt to the 8 XCDs round-robin by launch id.  This gives XCD x a CONTIGUOUS range of work ids, so that the
// workgroups sharing an operand tile (the column tiles of one row tile; the dW tiles of one row slab) are neighbours in ONE
// XCD's L2 instead of eight (bijective for any total: the first total % 8 XCDs take one more).
__device__ __forceinline__ unsigned tg_xcd_contiguous(unsigned launch_id, unsigned total) {
    const unsigned q = total >> 3, r = total & 7u, xcd = launch_id & 7u, i = launch_id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

struct TgNtArgs {
    const unsigned short *X; long long ldx;     // (R, K) bf16, row stride ldx elements (multiple of 8)
    const unsigned short *W; long long ldw;     // (N, K) bf16
    unsigned short *Y; long long ldy;           // (R, N) bf16
    const float *bias;                          // (N) fp32 or null: y = bf16(acc + bf16(bias))
    float *stats;                               // null, or [slots][N][2]: sum y, sum y^2 of the rounded outputs
    const float *xf;                            // null, or (4, K) fp32 [mean | invstd | scale | beta]: X is read through BatchNorm + ReLU
    // XF == 2 (data gradient straight behind a BatchNorm + ReLU backward): X holds dZ, the gradient of relu(bn(Yp)); the operand is
    // dY = scale (dZ [bn(Yp) > 0] - p - (Yp - mean) q) formed while it is staged (tg_bn_bwd8) and, by column tile 0, written to Xo
    const unsigned short *Xa; long long ldxa;   // Yp (R, K) bf16: the BatchNorm's input
    unsigned short *Xo; long long ldxo;         // dY (R, K) bf16 out
    const float *gf;                            // (4, K) fp32 [dgamma | dbeta | p | q] of that BatchNorm's backward
    // BS (the product IS the gradient of relu(bn(Bx)), e.g. the data gradient of the layer behind that BatchNorm): the statistics of
    // that BatchNorm's backward are taken from the rounded outputs in the epilogue — per slot and column sum g and sum g xhat with
    // g = y [bn(Bx) > 0], xhat = (Bx - mean) invstd: the two sums bn_cl_reduce_kernel<MODE 1> (bn_relu.hip) forms, element for element
    const unsigned short *Bx; long long ldbx;   // (R, N) bf16: that BatchNorm's input
    const float *bcf;                           // (4, N) fp32 [mean | invstd | scale | shift] (pdm_bn_finalize_stats)
    float *bstats;                              // [slots][N][2]
    // PL (the product is the input of BatchNorm + ReLU + max over the ns neighbours of a group — the tail of an SA scale, rows g ns ..
    // g ns + ns - 1 = group g): the epilogue also leaves every group's max / min of the ROUNDED outputs per channel with the first
    // index attaining them, (R / ns, N) each — what bn_pool_stats_kernel (bn_relu.hip) reads the whole tensor again for
    int pl_ns;
    unsigned short *pl_max, *pl_min;            // (R / ns, N) bf16
    unsigned char *pl_imax, *pl_imin;           // (R / ns, N)
    long long R;
    int K, N;
};

// BatchNorm(train) + ReLU applied to eight bf16 activations on their way into LDS: bf16(relu((x - mean) * scale + beta)) — the
// arithmetic of bn_cl_apply_kernel (bn_relu.hip), so the operand a layer reads this way is bit for bit the tensor that kernel
// would have written.  `live` = the row exists (rows past R must stay zero).
constexpr int TG_XFK = 512;      // widest BatchNorm the contractions apply on the fly (coefficient table in LDS)
struct TgBnCoef { float mu[8], sc[8], sh[8]; };
// a thread's eight channels k .. k + 7 are the same for every chunk it stages in a k-step: fetched once per step
__device__ __forceinline__ TgBnCoef tg_bn_coef8(const float *__restrict__ coef, int K, int k) {
    const float4 m0 = *reinterpret_cast<const float4 *>(coef + k), m1 = *reinterpret_cast<const float4 *>(coef + k + 4);
    const float4 s0 = *reinterpret_cast<const float4 *>(coef + 2 * K + k), s1 = *reinterpret_cast<const float4 *>(coef + 2 * K + k + 4);
    const float4 b0 = *reinterpret_cast<const float4 *>(coef + 3 * K + k), b1 = *reinterpret_cast<const float4 *>(coef + 3 * K + k + 4);
    return TgBnCoef{{m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w},
                    {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}};
}
__device__ __forceinline__ uint4 tg_bn_relu8(uint4 v, const TgBnCoef &c, bool live) {
    if (!live) return make_uint4(0, 0, 0, 0);
    const float *mu = c.mu, *sc = c.sc, *sh = c.sh;
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = fmaxf(fmaf(__uint_as_float(w[e] << 16) - mu[2 * e], sc[2 * e], sh[2 * e]), 0.f);
        const float hi = fmaxf(fmaf(__uint_as_float(w[e] & 0xffff0000u) - mu[2 * e + 1], sc[2 * e + 1], sh[2 * e + 1]), 0.f);
        o[e] = tg_pack2(lo, hi);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// BatchNorm(train) + ReLU BACKWARD applied to eight gradient values on their way into LDS: the arithmetic of
// bn_cl_apply_kernel<MODE 1> (bn_relu.hip), d = y - mean, g = dz [d scale + beta > 0], out = bf16(scale ((g - p) - d q)) — bit
// for bit the tensor that kernel would have written.
struct TgBnBwd { float mu[8], sc[8], sh[8], p[8], q[8]; };
__device__ __forceinline__ void tg_ld8(const float *__restrict__ src, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ TgBnBwd tg_bn_bwd_coef8(const float *__restrict__ coef, const float *__restrict__ grads, int K, int k) {
    TgBnBwd c;
    tg_ld8(coef + k, c.mu); tg_ld8(coef + 2 * K + k, c.sc); tg_ld8(coef + 3 * K + k, c.sh);
    tg_ld8(grads + 2 * K + k, c.p); tg_ld8(grads + 3 * K + k, c.q);
    return c;
}
__device__ __forceinline__ uint4 tg_bn_bwd8(uint4 dz, uint4 y, const TgBnBwd &c, bool live) {
    if (!live) return make_uint4(0, 0, 0, 0);
    const unsigned wz[4] = {dz.x, dz.y, dz.z, dz.w}, wy[4] = {y.x, y.y, y.z, y.w};
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * e + h;
            const float yv = h ? __uint_as_float(wy[e] & 0xffff0000u) : __uint_as_float(wy[e] << 16);
            const float gv = h ? __uint_as_float(wz[e] & 0xffff0000u) : __uint_as_float(wz[e] << 16);
            const float d = yv - c.mu[i];
            const float g = fmaf(d, c.sc[i], c.sh[i]) > 0.f ? gv : 0.f;
            r[h] = c.sc[i] * (g - c.p[i] - d * c.q[i]);
        }
        o[e] = tg_pack2(r[0], r[1]);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// Epilogue statistics of a BatchNorm + ReLU backward (TgNtArgs::bstats): eight rounded outputs `v` (the gradient of relu(bn(x)))
// and the eight inputs `xv` of that BatchNorm at the same place.  a += g, b = fma(g, d invstd, b) with d = x - mean,
// g = y [fma(d, scale, shift) > 0] — bn_cl_reduce_kernel<MODE 1>'s arithmetic.
struct TgBsCoef { float mu[8], is[8], sc[8], sh[8]; };
__device__ __forceinline__ void tg_bs_accum(uint4 v, uint4 xv, const TgBsCoef &c, float (&s1)[8], float (&s2)[8]) {
    const unsigned wy[4] = {v.x, v.y, v.z, v.w}, wx[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * e + h;
            const float yv = h ? __uint_as_float(wy[e] & 0xffff0000u) : __uint_as_float(wy[e] << 16);
            const float xx = h ? __uint_as_float(wx[e] & 0xffff0000u) : __uint_as_float(wx[e] << 16);
            const float d = xx - c.mu[i];
            const float g = fmaf(d, c.sc[i], c.sh[i]) > 0.f ? yv : 0.f;
            s1[i] += g;
            s2[i] = fmaf(g, d * c.is[i], s2[i]);
        }
    }
}
// the coefficients of columns n .. n + 7; a chunk beyond N reads those of columns 0 .. 7 (its outputs and inputs are zeros: g = 0)
__device__ __forceinline__ void tg_bs_coef8(TgBsCoef &c, const float *__restrict__ coef, int N, int n) {
    const int k = n < N ? n : 0;
    tg_ld8(coef + k, c.mu); tg_ld8(coef + N + k, c.is); tg_ld8(coef + 2 * N + k, c.sc); tg_ld8(coef + 3 * N + k, c.sh);
}

// 16-byte chunk `chunk` (0..7) of row `row` of a [rows][64] bf16 LDS tile: rows 2i, 2i+1 sit in the two 128-byte halves of
// a 256-byte bank line, the pair index permutes the chunk — 16 distinct rows reading one logical chunk hit 16 different
// (half, chunk) slots
__device__ __forceinline__ int tg_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Pooled-operator epilogue (TgNtArgs::pl_*): the groups of a row tile that sits in LDS as [BM][pitch] bf16.  Work item = (group of the
// tile, 16-byte chunk of 8 channels, QUARTER of the group's ns rows): four neighbouring lanes share a (group, chunk) and are folded with
// two shuffles — with one thread per (group, chunk) a quarter of the workgroup walked 16-32 rows each while the rest waited (+30 % on
// the widest-row contractions).  Strict comparisons keep the FIRST neighbour attaining an extreme (bn_pool_stats_kernel's rule); between
// quarters a tie goes to the lower quarter.
template <int BM, int CH, int YP>
__device__ __forceinline__ void tg_pool_tile(const TgNtArgs &a, const unsigned char *tile, long long row0, int col0, int t) {
    const int ns = a.pl_ns, gpt = BM / ns, nq = ns >> 2;          // host: ns a power of two, 4 <= ns <= 128
    for (int item = t; item < gpt * CH * 4; item += TG_T) {       // TG_T and the bound are multiples of 4: a quad stays together
        const int q = item & 3, gc = item >> 2;
        const int gi = gc / CH, ch = gc - gi * CH;
        const long long g = row0 / ns + gi;
        const int n = col0 + ch * 8;
        float mx[8], mn[8];
        unsigned ix[8], in_[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; mn[e] = INFINITY; ix[e] = in_[e] = 0u; }
        const unsigned char *p = tile + (size_t)(gi * ns + q * nq) * YP + ch * 16;
        for (int s_ = 0; s_ < nq; ++s_) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + (size_t)s_ * YP);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
            const unsigned si = (unsigned)(q * nq + s_);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                if (lo > mx[2 * e]) { mx[2 * e] = lo; ix[2 * e] = si; }
                if (lo < mn[2 * e]) { mn[2 * e] = lo; in_[2 * e] = si; }
                if (hi > mx[2 * e + 1]) { mx[2 * e + 1] = hi; ix[2 * e + 1] = si; }
                if (hi < mn[2 * e + 1]) { mn[2 * e + 1] = hi; in_[2 * e + 1] = si; }
            }
        }
#pragma unroll
        for (int off = 1; off <= 2; off <<= 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float omx = __shfl_xor(mx[e], off, 64), omn = __shfl_xor(mn[e], off, 64);
                const unsigned oix = __shfl_xor(ix[e], off, 64), oin = __shfl_xor(in_[e], off, 64);
                if (omx > mx[e] || (omx == mx[e] && oix < ix[e])) { mx[e] = omx; ix[e] = oix; }
                if (omn < mn[e] || (omn == mn[e] && oin < in_[e])) { mn[e] = omn; in_[e] = oin; }
            }
        }
        if (q != 0 || g * ns >= a.R || n >= a.N) continue;        // (after the shuffles: every lane of the quad took part)
        uint4 pm, pn;      // the extremes ARE bf16 values: their upper halves
        pm.x = (__float_as_uint(mx[0]) >> 16) | (__float_as_uint(mx[1]) & 0xffff0000u); pm.y = (__float_as_uint(mx[2]) >> 16) | (__float_as_uint(mx[3]) & 0xffff0000u);
        pm.z = (__float_as_uint(mx[4]) >> 16) | (__float_as_uint(mx[5]) & 0xffff0000u); pm.w = (__float_as_uint(mx[6]) >> 16) | (__float_as_uint(mx[7]) & 0xffff0000u);
        pn.x = (__float_as_uint(mn[0]) >> 16) | (__float_as_uint(mn[1]) & 0xffff0000u); pn.y = (__float_as_uint(mn[2]) >> 16) | (__float_as_uint(mn[3]) & 0xffff0000u);
        pn.z = (__float_as_uint(mn[4]) >> 16) | (__float_as_uint(mn[5]) & 0xffff0000u); pn.w = (__float_as_uint(mn[6]) >> 16) | (__float_as_uint(mn[7]) & 0xffff0000u);
        const size_t o = (size_t)g * a.N + n;
        *reinterpret_cast<uint4 *>(a.pl_max + o) = pm;
        *reinterpret_cast<uint4 *>(a.pl_min + o) = pn;
        uint2 bi, bn_;
        bi.x = ix[0] | (ix[1] << 8) | (ix[2] << 16) | (ix[3] << 24); bi.y = ix[4] | (ix[5] << 8) | (ix[6] << 16) | (ix[7] << 24);
        bn_.x = in_[0] | (in_[1] << 8) | (in_[2] << 16) | (in_[3] << 24); bn_.y = in_[4] | (in_[5] << 8) | (in_[6] << 16) | (in_[7] << 24);
        *reinterpret_cast<uint2 *>(a.pl_imax + o) = bi;
        *reinterpret_cast<uint2 *>(a.pl_imin + o) = bn_;
    }
}

// Tile shapes (4 waves, each 64 rows x JT * 32 columns): WN x JT = 2 x 2 -> 128 rows x 128 columns (wide layers),
// 1 x 2 -> 256 x 64, 1 x 1 -> 256 x 32 (the narrow first SA levels: no MFMA work and no LDS traffic on absent columns,
// twice the rows per workgroup behind one latency chain).
// PERSISTENT over row tiles: workgroup (slot, column tile) walks row tiles slot, slot + slots, ...; the next tile's first
// loads are requested before this tile's epilogue, the weights of a single-step contraction (K <= 64) are staged once, and
// the BatchNorm sums stay in registers until the end — one partial per slot (<= 1024: what the BatchNorm finalize kernel
// folds itself), no per-tile partials.
// The product is formed TRANSPOSED (D = W_tile . X_tile^T): a lane then owns one output ROW and, per 4 accumulator
// registers, 4 CONSECUTIVE channels — 8 bytes of bf16, one ds_write_b64 — where the direct form had 16 scattered 2-byte LDS
// writes per 32 x 32 tile (the epilogue's LDS writes alone were 1.7x the HBM time of a narrow tile).
template <int WN, int JT, int XF, bool BS = false, bool PL = false>
__global__ __launch_bounds__(TG_T, 2) void tg_nt_kernel(TgNtArgs a, int slots) {
    constexpr int BM = (4 / WN) * 64, BN = WN * JT * 32;
    constexpr int XB = BM * 128, WB = BN * 128;                  // bytes of the X / W stage (64 k x 2 B rows)
    constexpr int XI = BM * 8 / TG_T, WI = (BN * 8 + TG_T - 1) / TG_T;   // 16-byte chunks per thread
    constexpr int CH = BN / 8;                                   // 16-byte chunks per output row
    constexpr int YP = BN * 2 + 16;                              // pitch of the epilogue tile (8-byte writes down a column of rows)
    constexpr int YB = BM * YP;
    constexpr int MAIN = (XB + WB) > YB ? (XB + WB) : YB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN + 4 * BN * 2 * 4];
    unsigned char *Xs = smem, *Ws = smem + XB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int ncol = (a.N + BN - 1) / BN;
    const unsigned wg = ncol > 1 ? tg_xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;   // column tiles of a slot: one XCD
    const int col_tile = wg % ncol;
    const int slot = wg / ncol;
    const int col0 = col_tile * BN;
    const int nk = (a.K + TG_BK - 1) / TG_BK;
    const long long row_tiles = (a.R + BM - 1) / BM;

    // XF: the per-channel coefficients of the BatchNorm sit in LDS for the whole launch (K <= TG_XFK, host check): read from
    // global memory at the point of use they put an L2 round trip between the arrival of a stage's data and its LDS store
    __shared__ __attribute__((aligned(16))) float cfs[XF == 2 ? 6 * TG_XFK : XF == 1 ? 4 * TG_XFK : 4];   // rows of TG_XFK: mean, -, scale, shift [, p, q]
    if constexpr (XF != 0) {
        for (int c = threadIdx.x; c < a.K; c += TG_T) {
            cfs[c] = a.xf[c]; cfs[2 * TG_XFK + c] = a.xf[2 * a.K + c]; cfs[3 * TG_XFK + c] = a.xf[3 * a.K + c];
            if constexpr (XF == 2) { cfs[4 * TG_XFK + c] = a.gf[2 * a.K + c]; cfs[5 * TG_XFK + c] = a.gf[3 * a.K + c]; }
        }
        __syncthreads();
    }
    uint4 xr[XI], wr[WI], xa[XF == 2 ? XI : 1];
    auto load_x = [&](long long row0, int kt) {
        const int k0 = kt * TG_BK;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
            const int k = k0 + chunk * 8;
            const long long r = row0 + row;
            xr[i] = (r < a.R && k < a.K) ? *reinterpret_cast<const uint4 *>(a.X + r * a.ldx + k) : make_uint4(0, 0, 0, 0);
            if constexpr (XF == 2) xa[i] = (r < a.R && k < a.K) ? *reinterpret_cast<const uint4 *>(a.Xa + r * a.ldxa + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto load_w = [&](int kt) {
        const int k0 = kt * TG_BK;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
            const int k = k0 + chunk * 8;
            const int n = col0 + row;
            wr[i] = (row < BN && n < a.N && k < a.K) ? *reinterpret_cast<const uint4 *>(a.W + (long long)n * a.ldw + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
            if (row < BN) *reinterpret_cast<uint4 *>(Ws + tg_off(row, chunk)) = wr[i];
        }
    };
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    constexpr int RPP = TG_T / CH;               // rows per pass of the workgroup in the store phase
    const int chunk_o = t % CH;

    long long rt = slot;
    if (rt < row_tiles) { load_x(rt * BM, 0); load_w(0); }
    bool w_resident = false;                     // nk == 1: the weights stay in LDS across row tiles
    for (; rt < row_tiles; rt += slots) {
        const long long row0 = rt * BM;
        tg_f32x16 acc[2][JT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (XF == 2) {   // BatchNorm + ReLU backward on the way in; column tile 0 also writes the formed gradient out
                // two channels at a time over all of the thread's chunks (they share its eight channels): ten coefficient
                // registers live instead of forty
                const int k = kt * TG_BK + (t & 7) * 8, kc = k < a.K ? k : 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 mu = *reinterpret_cast<const float2 *>(cfs + kc + 2 * e), sc = *reinterpret_cast<const float2 *>(cfs + 2 * TG_XFK + kc + 2 * e),
                                 sh = *reinterpret_cast<const float2 *>(cfs + 3 * TG_XFK + kc + 2 * e), pp = *reinterpret_cast<const float2 *>(cfs + 4 * TG_XFK + kc + 2 * e),
                                 qq = *reinterpret_cast<const float2 *>(cfs + 5 * TG_XFK + kc + 2 * e);
#pragma unroll
                    for (int i = 0; i < XI; ++i) {
                        const unsigned wz = e == 0 ? xr[i].x : e == 1 ? xr[i].y : e == 2 ? xr[i].z : xr[i].w;
                        const unsigned wy = e == 0 ? xa[i].x : e == 1 ? xa[i].y : e == 2 ? xa[i].z : xa[i].w;
                        const float d0 = __uint_as_float(wy << 16) - mu.x, d1 = __uint_as_float(wy & 0xffff0000u) - mu.y;
                        const float g0 = fmaf(d0, sc.x, sh.x) > 0.f ? __uint_as_float(wz << 16) : 0.f;
                        const float g1 = fmaf(d1, sc.y, sh.y) > 0.f ? __uint_as_float(wz & 0xffff0000u) : 0.f;
                        const unsigned o = tg_pack2(sc.x * (g0 - pp.x - d0 * qq.x), sc.y * (g1 - pp.y - d1 * qq.y));
                        if (e == 0) xr[i].x = o; else if (e == 1) xr[i].y = o; else if (e == 2) xr[i].z = o; else xr[i].w = o;
                    }
                }
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
                    const bool live = row0 + row < a.R && k < a.K;
                    const uint4 v = live ? xr[i] : make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4 *>(Xs + tg_off(row, chunk)) = v;
                    if (col_tile == 0 && live) *reinterpret_cast<uint4 *>(a.Xo + (row0 + row) * a.ldxo + k) = v;
                }
            } else if constexpr (XF == 1) {   // the producer's BatchNorm + ReLU, applied here instead of in a pass of its own
                const int k = kt * TG_BK + (t & 7) * 8;      // chunk = (t + 256 i) & 7 = t & 7 for every i
                const TgBnCoef cf = tg_bn_coef8(cfs, TG_XFK, k < a.K ? k : 0);   // the (4, TG_XFK) table in LDS
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
                    *reinterpret_cast<uint4 *>(Xs + tg_off(row, chunk)) = tg_bn_relu8(xr[i], cf, row0 + row < a.R && k < a.K);
                }
            } else {
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
                    *reinterpret_cast<uint4 *>(Xs + tg_off(row, chunk)) = xr[i];
                }
            }
            if (!w_resident) store_w();
            __syncthreads();
            // the next loads travel behind the MFMAs (and, at the last step, behind this tile's epilogue)
            if (kt + 1 < nk) { load_x(row0, kt + 1); load_w(kt + 1); }
            else if (rt + slots < row_tiles) { load_x((rt + slots) * BM, 0); if (nk > 1) load_w(0); }
            const int rem = a.K - kt * TG_BK;
            const int ksteps = rem >= TG_BK ? TG_BK / 16 : (rem + 15) / 16;   // 16-deep steps that hold data (K = 8: one, not four)
            for (int s = 0; s < ksteps; ++s) {
                const int chunk = 2 * s + (lane >> 5);
                tg_bf16x8 af[2], bf[JT];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = wm * 64 + i * 32 + (lane & 31);
                    af[i] = __builtin_bit_cast(tg_bf16x8, *reinterpret_cast<const uint4 *>(Xs + tg_off(row, chunk)));
                }
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const int n = wn * JT * 32 + j * 32 + (lane & 31);
                    bf[j] = __builtin_bit_cast(tg_bf16x8, *reinterpret_cast<const uint4 *>(Ws + tg_off(n, chunk)));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < JT; ++j)   // D[n][r] = sum_k W[n][k] X[r][k]: the weights as the A operand
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
        // ---- epilogue: bf16 (RNE) into an LDS tile [BM][pitch YP]; lane = output row, 4 registers = 4 consecutive channels
        if (nk == 1 && XB + WB <= MAIN && YB <= XB) w_resident = true;      // the epilogue tile does not reach the weight stage
        TgBsCoef bcf;
        if constexpr (BS) {   // fetched per tile (L2-hot): held across the k-loop they cost the widest tile 20 spilled registers
            const float *cp = a.bcf;
            asm volatile("" : "+s"(cp));
            tg_bs_coef8(bcf, cp, a.N, col0 + chunk_o * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = wn * JT * 32 + j * 32 + 8 * g + 4 * (lane >> 5);
                    float b4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (XF != 2 && a.bias) {   // only the heads' last layers carry one: fetched here (L2) rather than held in 16 JT registers
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (col0 + col + e < a.N) b4[e] = tg_f32(tg_bf16(a.bias[col0 + col + e]));
                    }
                    uint2 v;
                    v.x = tg_pack2(acc[i][j][4 * g] + b4[0], acc[i][j][4 * g + 1] + b4[1]);
                    v.y = tg_pack2(acc[i][j][4 * g + 2] + b4[2], acc[i][j][4 * g + 3] + b4[3]);
                    *reinterpret_cast<uint2 *>(smem + row * YP + col * 2) = v;
                }
        }
        __syncthreads();
        if constexpr (PL) tg_pool_tile<BM, CH, YP>(a, smem, row0, col0, t);
        uint4 bx[BS ? BM / RPP : 1];
        if constexpr (BS) {   // the BatchNorm's inputs under this thread's output chunks, all requested at once (the accumulators are dead)
            const int n = col0 + chunk_o * 8;
#pragma unroll
            for (int i = 0; i < BM / RPP; ++i) {
                const long long r = row0 + t / CH + RPP * i;
                bx[i] = (r < a.R && n < a.N) ? *reinterpret_cast<const uint4 *>(a.Bx + r * a.ldbx + n) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < BM / RPP; ++i) {
            const int row = t / CH + RPP * i;
            const uint4 v = *reinterpret_cast<const uint4 *>(smem + row * YP + chunk_o * 16);
            const long long r = row0 + row;
            const int n = col0 + chunk_o * 8;
            if (r < a.R && n < a.N) *reinterpret_cast<uint4 *>(a.Y + r * a.ldy + n) = v;
            if constexpr (BS) {   // rows beyond R / columns beyond N hold zeros: g = 0
                tg_bs_accum(v, bx[i], bcf, s1, s2);
            }
            if (!BS && XF != 2 && a.stats) {   // rows beyond R were staged as zeros: they add nothing
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                    s1[2 * e] += lo; s2[2 * e] = fmaf(lo, lo, s2[2 * e]);
                    s1[2 * e + 1] += hi; s2[2 * e + 1] = fmaf(hi, hi, s2[2 * e + 1]);
                }
            }
        }
        __syncthreads();
    }
    if (BS || (XF != 2 && a.stats)) {   // one partial per slot.  Lanes with equal t % CH hold the same columns.
        float *red = reinterpret_cast<float *>(smem + MAIN);       // [4 waves][BN columns] sums, then the same of squares
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int off = CH; off < 64; off <<= 1) {
                s1[e] += __shfl_xor(s1[e], off, 64);
                s2[e] += __shfl_xor(s2[e], off, 64);
            }
        }
        if (lane < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[wave * BN + chunk_o * 8 + e] = s1[e];
                red[4 * BN + wave * BN + chunk_o * 8 + e] = s2[e];
            }
        }
        __syncthreads();
        if (t < BN && col0 + t < a.N) {
            const float sa = ((red[t] + red[BN + t]) + red[2 * BN + t]) + red[3 * BN + t];
            const float sb = ((red[4 * BN + t] + red[5 * BN + t]) + red[6 * BN + t]) + red[7 * BN + t];
            float *o = (BS ? a.bstats : a.stats) + ((long long)slot * a.N + col0 + t) * 2;
            o[0] = sa; o[1] = sb;
        }
    }
}

// The same contraction for the WIDE tiles (128 x 128) without a second operand stream, as a walk over STAGES (one k-step of one row
// tile) with the X rows of two stages ahead in flight (two register sets that swap roles stage by stage): a wide layer has 4-8
// k-steps per tile, and one 16 KB stage per workgroup in flight left the kernel waiting for memory at 3.1-3.7 TB/s (head layers
// 172 -> 152 us).  The narrow tiles keep the loop above: this structure runs them 15-20 % slower (measured).
// (no BS form: with the gradient statistics' sixteen sums this structure spills ~100 registers; wide products with statistics
// take tg_nt_kernel<2, 2, 0, true>)
template <int XF, bool PL = false>
__global__ __launch_bounds__(TG_T, 2) void tg_nt_deep_kernel(TgNtArgs a, int slots) {
    constexpr int WN = 2, JT = 2;
    constexpr int BM = (4 / WN) * 64, BN = WN * JT * 32;
    constexpr int XB = BM * 128, WB = BN * 128;                  // bytes of the X / W stage (64 k x 2 B rows)
    constexpr int XI = BM * 8 / TG_T, WI = (BN * 8 + TG_T - 1) / TG_T;   // 16-byte chunks per thread
    constexpr int CH = BN / 8;                                   // 16-byte chunks per output row
    constexpr int YP = BN * 2 + 16;                              // pitch of the epilogue tile (8-byte writes down a column of rows)
    constexpr int YB = BM * YP;
    constexpr int MAIN = (XB + WB) > YB ? (XB + WB) : YB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN + 4 * BN * 2 * 4];
    unsigned char *Xs = smem, *Ws = smem + XB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int ncol = (a.N + BN - 1) / BN;
    const unsigned wg = ncol > 1 ? tg_xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;   // column tiles of a slot: one XCD
    const int col_tile = wg % ncol;
    const int slot = wg / ncol;
    const int col0 = col_tile * BN;
    const int nk = (a.K + TG_BK - 1) / TG_BK;
    const long long row_tiles = (a.R + BM - 1) / BM;

    // XF: the per-channel coefficients of the BatchNorm sit in LDS for the whole launch (K <= TG_XFK, host check): read from
    // global memory at the point of use they put an L2 round trip between the arrival of a stage's data and its LDS store
    __shared__ __attribute__((aligned(16))) float cfs[XF == 2 ? 6 * TG_XFK : XF == 1 ? 4 * TG_XFK : 4];   // rows of TG_XFK: mean, -, scale, shift [, p, q]
    if constexpr (XF != 0) {
        for (int c = threadIdx.x; c < a.K; c += TG_T) {
            cfs[c] = a.xf[c]; cfs[2 * TG_XFK + c] = a.xf[2 * a.K + c]; cfs[3 * TG_XFK + c] = a.xf[3 * a.K + c];
            if constexpr (XF == 2) { cfs[4 * TG_XFK + c] = a.gf[2 * a.K + c]; cfs[5 * TG_XFK + c] = a.gf[3 * a.K + c]; }
        }
        __syncthreads();
    }
    // DEEP: the X rows of TWO stages ahead are in flight (two register sets that swap roles stage by stage) on the wide tiles
    // without a second operand stream — a wide layer has 4-8 k-steps per tile, and one stage of 16 KB per workgroup in
    // flight left the kernel waiting for memory at 3.1-3.7 TB/s.
    constexpr bool DEEP = WN == 2 && JT == 2 && XF != 2;
    uint4 xr[XI], xq[DEEP ? XI : 1], wr[WI], xa[XF == 2 ? XI : 1];
    auto load_x = [&](uint4 (&xd)[XI], uint4 (&xad)[XF == 2 ? XI : 1], long long row0, int kt) {
        const int k0 = kt * TG_BK;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
            const int k = k0 + chunk * 8;
            const long long r = row0 + row;
            xd[i] = (r < a.R && k < a.K) ? *reinterpret_cast<const uint4 *>(a.X + r * a.ldx + k) : make_uint4(0, 0, 0, 0);
            if constexpr (XF == 2) xad[i] = (r < a.R && k < a.K) ? *reinterpret_cast<const uint4 *>(a.Xa + r * a.ldxa + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto load_w = [&](int kt) {
        const int k0 = kt * TG_BK;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
            const int k = k0 + chunk * 8;
            const int n = col0 + row;
            wr[i] = (row < BN && n < a.N && k < a.K) ? *reinterpret_cast<const uint4 *>(a.W + (long long)n * a.ldw + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
            if (row < BN) *reinterpret_cast<uint4 *>(Ws + tg_off(row, chunk)) = wr[i];
        }
    };
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    constexpr int RPP = TG_T / CH;               // rows per pass of the workgroup in the store phase
    const int chunk_o = t % CH;

    // a stage = one k-step of one row tile; the workgroup walks (row tile slot, slot + slots, ...) x (k-step 0 .. nk - 1)
    long long rt = slot;
    int kt = 0;
    auto after = [&](long long &r, int &k) { if (++k == nk) { k = 0; r += slots; } };
    if (rt < row_tiles) {
        load_x(xr, xa, rt * BM, 0); load_w(0);
        if constexpr (DEEP) {
            long long r1 = rt; int k1 = kt; after(r1, k1);
            if (r1 < row_tiles) load_x(xq, xa, r1 * BM, k1);
        }
    }
    bool w_resident = false;                     // nk == 1: the weights stay in LDS across row tiles
    tg_f32x16 acc[2][JT];
    // one stage, its X rows in `xc` (requested one stage ago, or two with DEEP)
    auto stage = [&](uint4 (&xc)[XI], uint4 (&xac)[XF == 2 ? XI : 1]) {
        const long long row0 = rt * BM;
        if (kt == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        if constexpr (XF == 2) {   // BatchNorm + ReLU backward on the way in; column tile 0 also writes the formed gradient out
            // two channels at a time over all of the thread's chunks (they share its eight channels): ten coefficient
            // registers live instead of forty
            const int k = kt * TG_BK + (t & 7) * 8, kc = k < a.K ? k : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 mu = *reinterpret_cast<const float2 *>(cfs + kc + 2 * e), sc = *reinterpret_cast<const float2 *>(cfs + 2 * TG_XFK + kc + 2 * e),
                             sh = *reinterpret_cast<const float2 *>(cfs + 3 * TG_XFK + kc + 2 * e), pp = *reinterpret_cast<const float2 *>(cfs + 4 * TG_XFK + kc + 2 * e),
                             qq = *reinterpret_cast<const float2 *>(cfs + 5 * TG_XFK + kc + 2 * e);
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    const unsigned wz = e == 0 ? xc[i].x : e == 1 ? xc[i].y : e == 2 ? xc[i].z : xc[i].w;
                    const unsigned wy = e == 0 ? xac[i].x : e == 1 ? xac[i].y : e == 2 ? xac[i].z : xac[i].w;
                    const float d0 = __uint_as_float(wy << 16) - mu.x, d1 = __uint_as_float(wy & 0xffff0000u) - mu.y;
                    const float g0 = fmaf(d0, sc.x, sh.x) > 0.f ? __uint_as_float(wz << 16) : 0.f;
                    const float g1 = fmaf(d1, sc.y, sh.y) > 0.f ? __uint_as_float(wz & 0xffff0000u) : 0.f;
                    const unsigned o = tg_pack2(sc.x * (g0 - pp.x - d0 * qq.x), sc.y * (g1 - pp.y - d1 * qq.y));
                    if (e == 0) xc[i].x = o; else if (e == 1) xc[i].y = o; else if (e == 2) xc[i].z = o; else xc[i].w = o;
                }
            }
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
                const bool live = row0 + row < a.R && k < a.K;
                const uint4 v = live ? xc[i] : make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4 *>(Xs + tg_off(row, chunk)) = v;
                if (col_tile == 0 && live) *reinterpret_cast<uint4 *>(a.Xo + (row0 + row) * a.ldxo + k) = v;
            }
        } else if constexpr (XF == 1) {   // the producer's BatchNorm + ReLU, applied here instead of in a pass of its own
            const int k = kt * TG_BK + (t & 7) * 8;      // chunk = (t + 256 i) & 7 = t & 7 for every i
            const TgBnCoef cf = tg_bn_coef8(cfs, TG_XFK, k < a.K ? k : 0);   // the (4, TG_XFK) table in LDS
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
                *reinterpret_cast<uint4 *>(Xs + tg_off(row, chunk)) = tg_bn_relu8(xc[i], cf, row0 + row < a.R && k < a.K);
            }
        } else {
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int q = t + TG_T * i, row = q >> 3, chunk = q & 7;
                *reinterpret_cast<uint4 *>(Xs + tg_off(row, chunk)) = xc[i];
            }
        }
        if (!w_resident) store_w();
        __syncthreads();
        // the next loads travel behind the MFMAs (and, at a tile's last step, behind its epilogue)
        long long r1 = rt; int k1 = kt; after(r1, k1);
        if constexpr (DEEP) {
            long long r2 = r1; int k2 = k1; after(r2, k2);
            if (r2 < row_tiles) load_x(xc, xac, r2 * BM, k2);
        } else {
            if (r1 < row_tiles) load_x(xc, xac, r1 * BM, k1);
        }
        if (r1 < row_tiles && (nk > 1 || false)) load_w(k1);
        const int rem = a.K - kt * TG_BK;
        const int ksteps = rem >= TG_BK ? TG_BK / 16 : (rem + 15) / 16;   // 16-deep steps that hold data (K = 8: one, not four)
        for (int s = 0; s < ksteps; ++s) {
            const int chunk = 2 * s + (lane >> 5);
            tg_bf16x8 af[2], bf[JT];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + (lane & 31);
                af[i] = __builtin_bit_cast(tg_bf16x8, *reinterpret_cast<const uint4 *>(Xs + tg_off(row, chunk)));
            }
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const int n = wn * JT * 32 + j * 32 + (lane & 31);
                bf[j] = __builtin_bit_cast(tg_bf16x8, *reinterpret_cast<const uint4 *>(Ws + tg_off(n, chunk)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j)   // D[n][r] = sum_k W[n][k] X[r][k]: the weights as the A operand
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt == nk - 1) {
            // ---- epilogue: bf16 (RNE) into an LDS tile [BM][pitch YP]; lane = output row, 4 registers = 4 consecutive channels
            if (nk == 1 && XB + WB <= MAIN && YB <= XB) w_resident = true;      // the epilogue tile does not reach the weight stage
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = wn * JT * 32 + j * 32 + 8 * g + 4 * (lane >> 5);
                        float b4[4] = {0.f, 0.f, 0.f, 0.f};
                        if (XF != 2 && a.bias) {   // only the heads' last layers carry one: fetched here (L2) rather than held in 16 JT registers
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col0 + col + e < a.N) b4[e] = tg_f32(tg_bf16(a.bias[col0 + col + e]));
                        }
                        uint2 v;
                        v.x = tg_pack2(acc[i][j][4 * g] + b4[0], acc[i][j][4 * g + 1] + b4[1]);
                        v.y = tg_pack2(acc[i][j][4 * g + 2] + b4[2], acc[i][j][4 * g + 3] + b4[3]);
                        *reinterpret_cast<uint2 *>(smem + row * YP + col * 2) = v;
                    }
            }
            __syncthreads();
            if constexpr (PL) tg_pool_tile<BM, CH, YP>(a, smem, row0, col0, t);
#pragma unroll
            for (int i = 0; i < BM / RPP; ++i) {
                const int row = t / CH + RPP * i;
                const uint4 v = *reinterpret_cast<const uint4 *>(smem + row * YP + chunk_o * 16);
                const long long r = row0 + row;
                const int n = col0 + chunk_o * 8;
                if (r < a.R && n < a.N) *reinterpret_cast<uint4 *>(a.Y + r * a.ldy + n) = v;
                if (XF != 2 && a.stats) {   // rows beyond R were staged as zeros: they add nothing
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                        s1[2 * e] += lo; s2[2 * e] = fmaf(lo, lo, s2[2 * e]);
                        s1[2 * e + 1] += hi; s2[2 * e + 1] = fmaf(hi, hi, s2[2 * e + 1]);
                    }
                }
            }
            __syncthreads();
        }
        rt = r1; kt = k1;
    };
    if constexpr (DEEP) {
        while (rt < row_tiles) {
            stage(xr, xa);
            if (!(rt < row_tiles)) break;
            stage(xq, xa);
        }
    } else {
        while (rt < row_tiles) stage(xr, xa);
    }
    if (XF != 2 && a.stats) {   // one partial per slot.  Lanes with equal t % CH hold the same columns.
        float *red = reinterpret_cast<float *>(smem + MAIN);       // [4 waves][BN columns] sums, then the same of squares
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int off = CH; off < 64; off <<= 1) {
                s1[e] += __shfl_xor(s1[e], off, 64);
                s2[e] += __shfl_xor(s2[e], off, 64);
            }
        }
        if (lane < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[wave * BN + chunk_o * 8 + e] = s1[e];
                red[4 * BN + wave * BN + chunk_o * 8 + e] = s2[e];
            }
        }
        __syncthreads();
        if (t < BN && col0 + t < a.N) {
            const float sa = ((red[t] + red[BN + t]) + red[2 * BN + t]) + red[3 * BN + t];
            const float sb = ((red[4 * BN + t] + red[5 * BN + t]) + red[6 * BN + t]) + red[7 * BN + t];
            float *o = a.stats + ((long long)slot * a.N + col0 + t) * 2;
            o[0] = sa; o[1] = sb;
        }
    }
}

// ---- weight gradient -------------------------------------------------------------------------------------------------
struct TgTnArgs {
    const unsigned short *dY; long long ldy;    // (R, N) bf16
    const unsigned short *X; long long ldx;     // (R, K) bf16
    float *partial;                             // [slabs][N][K] fp32
    const float *xf;                            // null, or (4, K) fp32: X is read through BatchNorm + ReLU (see tg_bn_relu8)
    long long R, rows_per_slab;
    int N, K;
};

// Operand fragment of v_mfma_f32_32x32x16_bf16 for a contraction over the ROWS of a row-major LDS tile: lane l needs
// rows r0 + 8 (l >> 5) + j, j = 0..7, of column c0 + (l & 31).  ds_read_b64_tr_b16 hands a 16-lane group a 4-row x 16-column
// block transposed: lane 4q + p of the group supplies the address of (row q, columns 4p..4p+3), lane i receives column i
// of the four rows.  Two such reads (rows +0..3, +4..7) make the eight elements.
__device__ __forceinline__ tg_bf16x8 tg_tr_frag(const unsigned char *tile, int r0, int c0, int lane) {
    const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3, h = g >> 1;
    const unsigned char *p0 = tile + (r0 + 8 * h + q) * TG_WPITCH + (c0 + 16 * (g & 1) + 4 * p) * 2;
    typedef __attribute__((address_space(3))) tg_s16x4 *lds_ptr;
    const tg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
    const tg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0 + 4 * TG_WPITCH));
    tg_s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(tg_bf16x8, v);
}

template <bool XF>
__global__ __launch_bounds__(TG_T) void tg_tn_kernel(TgTnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TG_WR * TG_WPITCH];
    unsigned char *Gs = smem, *Xs = smem + TG_WR * TG_WPITCH;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nkt = (a.K + 127) / 128;
    // one-dimensional grid of tiles x slabs; the dW tiles of one slab (they re-read its dY and X rows) are neighbours in one XCD
    const unsigned ntile = (unsigned)(((a.N + 127) / 128) * nkt);
    const unsigned wg = ntile > 1 ? tg_xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    const unsigned tile = wg % ntile;
    const int n0 = (tile / nkt) * 128, k0 = (tile % nkt) * 128;
    const long long slab = wg / ntile;
    const long long r_begin = slab * a.rows_per_slab;
    const long long r_end = r_begin + a.rows_per_slab < a.R ? r_begin + a.rows_per_slab : a.R;

    uint4 gr[4], xr[4];
    auto load = [&](long long r0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = t + TG_T * i, row = q >> 4, chunk = q & 15;
            const long long r = r0 + row;
            const int n = n0 + chunk * 8, k = k0 + chunk * 8;
            gr[i] = (r < r_end && n < a.N) ? *reinterpret_cast<const uint4 *>(a.dY + r * a.ldy + n) : make_uint4(0, 0, 0, 0);
            xr[i] = (r < r_end && k < a.K) ? *reinterpret_cast<const uint4 *>(a.X + r * a.ldx + k) : make_uint4(0, 0, 0, 0);
        }
    };
    tg_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // BatchNorm + ReLU on the X operand: this thread's eight channels k0 + 8 (t & 15) .. are the same for every stage
    TgBnCoef xcf{};
    if constexpr (XF) { const int k = k0 + (t & 15) * 8; xcf = tg_bn_coef8(a.xf, a.K, k < a.K ? k : 0); }
    if (r_begin < r_end) load(r_begin);
    for (long long r0 = r_begin; r0 < r_end; r0 += TG_WR) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = t + TG_T * i, row = q >> 4, chunk = q & 15;
            *reinterpret_cast<uint4 *>(Gs + row * TG_WPITCH + chunk * 16) = gr[i];
            uint4 v = xr[i];
            if constexpr (XF) v = tg_bn_relu8(v, xcf, r0 + row < r_end && k0 + chunk * 8 < a.K);
            *reinterpret_cast<uint4 *>(Xs + row * TG_WPITCH + chunk * 16) = v;
        }
        __syncthreads();
        if (r0 + TG_WR < r_end) load(r0 + TG_WR);
#pragma unroll
        for (int rc = 0; rc < TG_WR / 16; ++rc) {
            tg_bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = tg_tr_frag(Gs, rc * 16, wm * 64 + i * 32, lane);     // A[n][r] = dY[r][n]
                bf[i] = tg_tr_frag(Xs, rc * 16, wn * 64 + i * 32, lane);     // B[r][k] = X[r][k]
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    float *P = a.partial + slab * (long long)a.N * a.K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < a.N && k < a.K) P[(long long)n * a.K + k] = acc[i][j][r];
            }
        }
}

// Weight gradient of NARROW layers (N, K <= C, C = 32 or 64: the first SA levels, 4 M rows each).  The 128 x 128 tile above
// ran 16 x the MFMA work a 32 x 32 result needs and was bound by it (SQ_VALU_MFMA_BUSY 134 M cycles on a 200 us launch).
// Here the WAVES split the ROWS of a stage: a stage holds WR rows of dY and X (C columns each, row-major as in memory), wave w
// takes the 16-row chunks w, w + 4, ... and keeps the whole C x C result in its accumulators; the four partial results are
// added in wave order at the end (fixed order: reproducible).  C = 32: 64-byte LDS rows are conflict-free for the transposing
// reads as they are; C = 64: pitch 192 bytes.
template <int C, bool XF>
__global__ __launch_bounds__(TG_T) void tg_tn_narrow_kernel(TgTnArgs a) {
    constexpr int WR = C == 32 ? 256 : 128;                  // rows per stage
    constexpr int P = C == 32 ? 64 : 192;                    // LDS row pitch in bytes
    constexpr int CPR = C / 8;                               // 16-byte chunks per row
    constexpr int T = C / 32;                                // 32 x 32 tiles per side
    constexpr int STAGE = WR * P;
    constexpr int SMEM = 2 * STAGE > C * C * 4 ? 2 * STAGE : C * C * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char *Gs = smem, *Xs = smem + STAGE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long slab = blockIdx.y;
    const long long r_begin = slab * a.rows_per_slab;
    const long long r_end = r_begin + a.rows_per_slab < a.R ? r_begin + a.rows_per_slab : a.R;
    constexpr int LI = WR * CPR / TG_T;                      // chunks per thread and operand (4)
    uint4 gr[LI], xr[LI];
    auto load = [&](long long r0) {
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            const int q = t + TG_T * i, row = q / CPR, chunk = q % CPR;
            const long long r = r0 + row;
            gr[i] = (r < r_end && chunk * 8 < a.N) ? *reinterpret_cast<const uint4 *>(a.dY + r * a.ldy + chunk * 8) : make_uint4(0, 0, 0, 0);
            xr[i] = (r < r_end && chunk * 8 < a.K) ? *reinterpret_cast<const uint4 *>(a.X + r * a.ldx + chunk * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    tg_f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    typedef __attribute__((address_space(3))) tg_s16x4 *lds_ptr;
    auto frag = [&](const unsigned char *tile, int r0, int c0) {   // tg_tr_frag with this kernel's pitch
        const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3, h = g >> 1;
        const unsigned char *p0 = tile + (r0 + 8 * h + q) * P + (c0 + 16 * (g & 1) + 4 * p) * 2;
        const tg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
        const tg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0 + 4 * P));
        tg_s16x8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(tg_bf16x8, v);
    };
    TgBnCoef xcf{};   // BatchNorm + ReLU on the X operand: this thread's eight channels are the same for every stage
    if constexpr (XF) { const int k = (t % CPR) * 8; xcf = tg_bn_coef8(a.xf, a.K, k < a.K ? k : 0); }
    if (r_begin < r_end) load(r_begin);
    for (long long r0 = r_begin; r0 < r_end; r0 += WR) {
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            const int q = t + TG_T * i, row = q / CPR, chunk = q % CPR;
            *reinterpret_cast<uint4 *>(Gs + row * P + chunk * 16) = gr[i];
            uint4 v = xr[i];
            if constexpr (XF) v = tg_bn_relu8(v, xcf, r0 + row < r_end && chunk * 8 < a.K);
            *reinterpret_cast<uint4 *>(Xs + row * P + chunk * 16) = v;
        }
        __syncthreads();
        if (r0 + WR < r_end) load(r0 + WR);
#pragma unroll
        for (int c = 0; c < WR / 64; ++c) {                  // this wave's 16-row chunks: wave, wave + 4, ...
            const int rc = (c * 4 + wave) * 16;
            tg_bf16x8 af[T], bf[T];
#pragma unroll
            for (int i = 0; i < T; ++i) { af[i] = frag(Gs, rc, i * 32); bf[i] = frag(Xs, rc, i * 32); }
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // the four waves' results, added in wave order through an LDS tile [C][C] fp32
    float *red = reinterpret_cast<float *>(smem);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), k = j * 32 + (lane & 31);
                        red[n * C + k] = w == 0 ? acc[i][j][r] : red[n * C + k] + acc[i][j][r];
                    }
        }
        __syncthreads();
    }
    float *Pout = a.partial + slab * (long long)a.N * a.K;
    for (int e = t; e < C * C; e += TG_T) {
        const int n = e / C, k = e % C;
        if (n < a.N && k < a.K) Pout[(long long)n * a.K + k] = red[e];
    }
}

// out[e] (+)= sum over k < parts of in[k][e], in ONE launch and in a fixed order: a workgroup owns 64 consecutive elements (256-byte
// lines of every part), its QG groups of 64 threads walk the parts q, q + QG, ... (eight loads in flight per thread, added in
// order), and the QG partial sums are added in group order through LDS.  (History: one thread per element walking thousands of
// slabs was a 200 us latency chain per layer; a tree of 32-way levels fixed that at two launches per weight gradient — 83 launches
// and 0.7 ms of a training step; this form is one launch for any number of parts.)
constexpr int TG_FOLD_QG = 16;
__global__ __launch_bounds__(64 * TG_FOLD_QG) void tg_fold_kernel(const float *__restrict__ in, int parts, long long elems, float *__restrict__ out,
                                                                  int accumulate) {
    __shared__ float red[TG_FOLD_QG][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, qg = blockDim.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (e < elems) {
        int k = q;
        for (; k + 7 * qg < parts; k += 8 * qg) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[(long long)(k + u * qg) * elems + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < parts; k += qg) s += in[(long long)k * elems + e];
    }
    red[q][lane] = s;
    __syncthreads();
    if (q == 0 && e < elems) {
        float tot = red[0][lane];
        for (int g = 1; g < qg; ++g) tot += red[g][lane];
        out[e] = accumulate ? out[e] + tot : tot;
    }
}

// folds in[parts][elems] down to out[elems] (accumulate: out += ...); `scratch` is unused (kept for the callers' workspace layout)
static int tg_fold(hipStream_t stream, const float *in, int parts, long long elems, float *out, int accumulate, float *scratch) {
    (void)scratch;
    int qg = 1;
    while (qg < TG_FOLD_QG && qg * 16 < parts) qg <<= 1;      // >= 16 parts per group before another group is added
    hipLaunchKernelGGL(tg_fold_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(64 * qg), 0, stream, in, parts, elems, out, accumulate);
    return check_launch("tg_fold");
}
static size_t tg_fold_scratch_floats(long long parts, long long elems) { (void)parts; (void)elems; return 0; }

// W (N, K) fp32 -> Wb (N, ldb) bf16 and / or Wt (K, ldt) bf16 (transposed), zero padded to the strides
// (rows_b rows of Wb and rows_t rows of Wt are written: rows >= N of Wb / >= K of Wt are zero padding too)
__global__ __launch_bounds__(256) void tg_pack_weight_kernel(const float *__restrict__ W, int N, int K, unsigned short *__restrict__ Wb, int ldb,
                                                             unsigned short *__restrict__ Wt, int ldt, int rows_b, int rows_t) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (Wb && e < (long long)rows_b * ldb) {
        const int n = (int)(e / ldb), k = (int)(e % ldb);
        Wb[e] = (n < N && k < K) ? tg_bf16(W[(long long)n * K + k]) : (unsigned short)0;
    }
    if (Wt && e < (long long)rows_t * ldt) {
        const int k = (int)(e / ldt), n = (int)(e % ldt);
        Wt[e] = (k < K && n < N) ? tg_bf16(W[(long long)n * K + k]) : (unsigned short)0;
    }
}

// The same for MANY layers in one launch: job j owns blocks first_block[j] .. first_block[j + 1] - 1 of the grid (a binary search
// over <= a few hundred jobs) and writes its (rows_to, cols_to) pair whole.  Table in device memory (pdm_tg_pack_weight_many).
struct TgPackJob { const float *W; unsigned short *Wb, *Wt; int N, K, rows_to, cols_to; long long first_block; };
__global__ __launch_bounds__(256) void tg_pack_weight_many_kernel(const TgPackJob *__restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                                    // last job whose first block <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TgPackJob j = jobs[lo];
    const long long e = ((long long)blockIdx.x - j.first_block) * 256 + threadIdx.x;
    if (e >= (long long)j.rows_to * j.cols_to) return;
    {
        const int n = (int)(e / j.cols_to), k = (int)(e % j.cols_to);
        j.Wb[e] = (n < j.N && k < j.K) ? tg_bf16(j.W[(long long)n * j.K + k]) : (unsigned short)0;
    }
    {
        const int k = (int)(e / j.rows_to), n = (int)(e % j.rows_to);
        j.Wt[e] = (k < j.K && n < j.N) ? tg_bf16(j.W[(long long)n * j.K + k]) : (unsigned short)0;
    }
}

// column sums of a (R, N) bf16 matrix in fp32 (the bias gradient of a Linear / convolution with bias): slot s walks rows
// s * rows_per_slot ..., a thread keeps 8 columns; partial[slot][N], folded by tg_fold.  CHP = chunks per row rounded up to a
// power of two <= 64.
__global__ __launch_bounds__(256) void tg_colsum_kernel(const unsigned short *__restrict__ Y, long long ld, long long R, int N, int chp,
                                                        long long rows_per_slot, float *__restrict__ partial) {
    __shared__ float red[4 * 512];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int chunk = t & (chp - 1), rpp = 256 / chp;
    const long long r0 = (long long)blockIdx.x * rows_per_slot;
    const long long r1 = r0 + rows_per_slot < R ? r0 + rows_per_slot : R;
    float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (chunk * 8 < N) {
        for (long long r = r0 + t / chp; r < r1; r += rpp) {
            const uint4 v = *reinterpret_cast<const uint4 *>(Y + r * ld + chunk * 8);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[2 * e] += __uint_as_float(w[e] << 16); s1[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
        for (int off = chp; off < 64; off <<= 1) s1[e] += __shfl_xor(s1[e], off, 64);
    if (lane < chp && chunk * 8 < N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * 512 + chunk * 8 + e] = s1[e];
    }
    __syncthreads();
    for (int c = t; c < N; c += 256)
        partial[(long long)blockIdx.x * N + c] = ((red[c] + red[512 + c]) + red[1024 + c]) + red[1536 + c];
}

static inline bool tg_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace pdm

using namespace pdm;

// tile shape by output width (see tg_nt_kernel): rows per workgroup
static inline int tg_bm_for(int N) { return N <= 64 ? 256 : 128; }
static inline int tg_bn_for(int N) { return N <= 32 ? 32 : N <= 64 ? 64 : 128; }
// persistent slots (workgroups per column tile): every slot walks row tiles slot, slot + slots, ...
static inline int tg_slots(long long rows, int N) {
    const long long row_tiles = (rows + tg_bm_for(N) - 1) / tg_bm_for(N);
    const int ncol = (N + tg_bn_for(N) - 1) / tg_bn_for(N);
    long long s = 1024 / ncol;                       // ~4 workgroups per CU in all
    if (s < 64) s = 64;
    if (s > 1024) s = 1024;
    return (int)(row_tiles < s ? row_tiles : s);
}

// statistics of pdm_tg_gemm_nt: (parts, N, 2) fp32, parts = pdm_tg_stats_parts(rows, N) <= 1024 — one partial per persistent slot
extern "C" int pdm_tg_stats_parts(long long rows, int N) { return rows <= 0 || N <= 0 ? 0 : tg_slots(rows, N); }

// Y (R, N) bf16 = X (R, K) bf16 . W (N, K)^T bf16 [+ bias], fp32 accumulation, one rounding.  Strides in elements, multiples
// of 8; K and N multiples of 8; pointers 16-byte aligned.  stats: null, or (pdm_tg_stats_parts(R, N), N, 2) fp32 = per slot the
// column sums of y and y^2 of the ROUNDED outputs (pdm_bn_relu_forward_stats folds them in double, in slot order).
// x_bn_coef: null, or (4, K) fp32 [mean | invstd | gamma invstd | beta] (pdm_bn_finalize_stats): X holds the PRE-BatchNorm
// outputs of the layer before and is read through bf16(relu((x - mean) scale + beta)) — that layer's BatchNorm + ReLU without a
// pass (and a tensor) of its own.
struct TgPool { int ns; void *xmax, *xmin; unsigned char *imax, *imin; };

static int tg_gemm_nt_impl(const char *who, void *stream, long long R, int K, int N, const void *X, long long ldx, const void *W, long long ldw,
                           void *Y, long long ldy, const float *bias, float *stats, const float *x_bn_coef, const void *Bx, long long ldbx,
                           const float *bcoef, float *bstats, const TgPool *pool = nullptr) {
    PDM_REQUIRE(R >= 0 && K >= 0 && N >= 0, PDM_E_BADARG, "%s: negative size", who);
    if (R == 0 || N == 0) return 0;
    PDM_REQUIRE(X && W && Y, PDM_E_BADARG, "%s: null pointer", who);
    PDM_REQUIRE(K % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0 && ldx >= K && ldw >= K && ldy >= N,
                PDM_E_BADARG, "%s: K=%d N=%d ldx=%lld ldw=%lld ldy=%lld must be multiples of 8 and cover the rows", who, K, N, ldx, ldw, ldy);
    PDM_REQUIRE(tg_al16(X) && tg_al16(W) && tg_al16(Y), PDM_E_BADARG, "%s: operands must be 16-byte aligned", who);
    PDM_REQUIRE(!x_bn_coef || K <= TG_XFK, PDM_E_TOOLARGE, "%s: x_bn_coef with K=%d (<= %d)", who, K, TG_XFK);
    const bool bs = bstats != nullptr;
    if (bs) {
        PDM_REQUIRE(Bx && bcoef && !stats && !x_bn_coef && !bias, PDM_E_BADARG, "%s: the gradient statistics take Bx and bcoef and exclude stats / x_bn_coef / bias", who);
        PDM_REQUIRE(ldbx % 8 == 0 && ldbx >= N && tg_al16(Bx) && tg_al16(bcoef), PDM_E_BADARG, "%s: ldbx=%lld (a multiple of 8, >= N), 16-byte aligned Bx / bcoef", who, ldbx);
    }
    const int bn = tg_bn_for(N);
    const int slots = tg_slots(R, N);
    const unsigned wgs = (unsigned)slots * (unsigned)((N + bn - 1) / bn);
    TgNtArgs a{};
    a.X = static_cast<const unsigned short *>(X); a.ldx = ldx; a.W = static_cast<const unsigned short *>(W); a.ldw = ldw;
    a.Y = static_cast<unsigned short *>(Y); a.ldy = ldy; a.bias = bias; a.stats = stats; a.xf = x_bn_coef; a.R = R; a.K = K; a.N = N;
    a.Bx = static_cast<const unsigned short *>(Bx); a.ldbx = ldbx; a.bcf = bcoef; a.bstats = bstats;
    if (pool) {
        PDM_REQUIRE(!bs && pool->ns >= 4 && pool->ns <= 128 && (pool->ns & (pool->ns - 1)) == 0 && R % pool->ns == 0, PDM_E_BADARG,
                    "%s: pooled groups of ns=%d rows (a power of two, 4 .. 128, dividing R=%lld)", who, pool->ns, R);
        PDM_REQUIRE(pool->xmax && pool->xmin && pool->imax && pool->imin && tg_al16(pool->xmax) && tg_al16(pool->xmin) &&
                    (reinterpret_cast<uintptr_t>(pool->imax) & 7u) == 0 && (reinterpret_cast<uintptr_t>(pool->imin) & 7u) == 0, PDM_E_BADARG,
                    "%s: null or misaligned pool outputs", who);
        a.pl_ns = pool->ns; a.pl_max = static_cast<unsigned short *>(pool->xmax); a.pl_min = static_cast<unsigned short *>(pool->xmin);
        a.pl_imax = pool->imax; a.pl_imin = pool->imin;
#define TG_NT_PL(WN, JT)                                                                                                              \
        do {                                                                                                                          \
            if (x_bn_coef) hipLaunchKernelGGL((tg_nt_kernel<WN, JT, 1, false, true>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);  \
            else hipLaunchKernelGGL((tg_nt_kernel<WN, JT, 0, false, true>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);   \
        } while (0)
        if (bn == 32) TG_NT_PL(1, 1);
        else if (bn == 64) TG_NT_PL(1, 2);
        else if (x_bn_coef) hipLaunchKernelGGL((tg_nt_deep_kernel<1, true>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);
        else hipLaunchKernelGGL((tg_nt_deep_kernel<0, true>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);
#undef TG_NT_PL
        return check_launch(who);
    }
#define TG_NT(WN, JT)                                                                                                        \
    do {                                                                                                                     \
        if (bs) hipLaunchKernelGGL((tg_nt_kernel<WN, JT, 0, true>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);   \
        else if (x_bn_coef) hipLaunchKernelGGL((tg_nt_kernel<WN, JT, 1>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);  \
        else hipLaunchKernelGGL((tg_nt_kernel<WN, JT, 0>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);           \
    } while (0)
    if (bn == 32) TG_NT(1, 1);
    else if (bn == 64) TG_NT(1, 2);
    else if (bs) hipLaunchKernelGGL((tg_nt_kernel<2, 2, 0, true>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);   // (the two-stages-ahead form has no registers left for the statistics)
    else if (x_bn_coef) hipLaunchKernelGGL((tg_nt_deep_kernel<1>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);
    else hipLaunchKernelGGL((tg_nt_deep_kernel<0>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);
#undef TG_NT
    return check_launch(who);
}

extern "C" int pdm_tg_gemm_nt(void *stream, long long R, int K, int N, const void *X, long long ldx, const void *W, long long ldw,
                              void *Y, long long ldy, const float *bias, float *stats, const float *x_bn_coef) {
    return tg_gemm_nt_impl("tg_gemm_nt", stream, R, K, N, X, ldx, W, ldw, Y, ldy, bias, stats, x_bn_coef, nullptr, 0, nullptr, nullptr);
}

// The same product when Y IS the gradient of relu(bn(Bx)) (the data gradient of the layer behind a BatchNorm + ReLU): the epilogue
// also leaves, per slot, the column sums of g = y [bn(Bx) > 0] and of g xhat over the ROUNDED outputs in bstats
// (pdm_tg_stats_parts(R, N), N, 2) — what pdm_bn_relu_backward_stats' reduce pass forms from a second read of Y and Bx
// (pdm_bn_finalize_bwd_stats folds the parts).  Bx (R, N) bf16 = that BatchNorm's input, bcoef (4, N) from pdm_bn_finalize_stats.
extern "C" int pdm_tg_gemm_nt_bs(void *stream, long long R, int K, int N, const void *X, long long ldx, const void *W, long long ldw,
                                 void *Y, long long ldy, const void *Bx, long long ldbx, const float *bcoef, float *bstats) {
    PDM_REQUIRE(bstats, PDM_E_BADARG, "tg_gemm_nt_bs: null pointer");
    return tg_gemm_nt_impl("tg_gemm_nt_bs", stream, R, K, N, X, ldx, W, ldw, Y, ldy, nullptr, nullptr, nullptr, Bx, ldbx, bcoef, bstats);
}

// Data gradient straight behind a BatchNorm + ReLU backward: dX (R, N) = dY (R, K) . W (N, K)^T where dY is not in memory yet:
//   dY = scale (dZ [bn(Yp) > 0] - p - (Yp - mean) q)        dZ, Yp (R, K) bf16; coef (4, K) as pdm_bn_finalize_stats leaves it;
//                                                          grads (4, K) as pdm_bn_relu_backward_stats leaves it
// is formed while the operand is staged and written to dYout (R, K) on the way (the weight gradient of the layer reads it from
// there): the values of pdm_bn_relu_backward's dx bit for bit, without that operator's pass over dZ and Yp and without this
// contraction's own read of dY.  Everything else as pdm_tg_gemm_nt (no bias, no statistics).
// pdm_tg_gemm_nt whose product is the input of the SA scales' tail, BatchNorm + ReLU + max over the ns neighbours of a group (rows
// g ns .. g ns + ns - 1 = group g; pointnet2_modules.py:46-52): besides Y and the BatchNorm column sums (`stats`), the epilogue leaves
// every group's max / min of the rounded outputs per channel and the first index attaining them — xmax, xmin (R / ns, N) bf16,
// imax, imin (R / ns, N) bytes, exactly what pdm_bn_relu_pool_forward's statistics pass computes from a second read of Y
// (pdm_bn_relu_pool_forward_kept finishes the operator from them).  ns a power of two, 4 .. 128, dividing R.
extern "C" int pdm_tg_gemm_nt_pool(void *stream, long long R, int K, int N, const void *X, long long ldx, const void *W, long long ldw,
                                   void *Y, long long ldy, const float *bias, float *stats, const float *x_bn_coef, int ns, void *xmax,
                                   void *xmin, unsigned char *imax, unsigned char *imin) {
    const TgPool pool{ns, xmax, xmin, imax, imin};
    return tg_gemm_nt_impl("tg_gemm_nt_pool", stream, R, K, N, X, ldx, W, ldw, Y, ldy, bias, stats, x_bn_coef, nullptr, 0, nullptr, nullptr, &pool);
}

static int tg_dy_slots(long long R, int N) {
    const int bn = N <= 64 ? 64 : 128, bm = 128;
    const long long row_tiles = (R + bm - 1) / bm;
    const int ncol = (N + bn - 1) / bn;
    long long sl = 1024 / ncol;
    if (sl < 64) sl = 64;
    if (sl > row_tiles) sl = row_tiles;
    return (int)sl;
}
// parts of pdm_tg_gemm_nt_dy_bs' statistics: (parts, N, 2) fp32
extern "C" int pdm_tg_dy_stats_parts(long long rows, int N) { return rows <= 0 || N <= 0 ? 0 : tg_dy_slots(rows, N); }

static int tg_gemm_nt_dy_impl(const char *who, void *stream, long long R, int K, int N, const void *dZ, long long lddz, const void *Yp, long long ldyp,
                              const void *W, long long ldw, void *dX, long long lddx, void *dYout, long long lddy, const float *coef,
                              const float *grads, const void *Bx, long long ldbx, const float *bcoef, float *bstats) {
    PDM_REQUIRE(R >= 0 && K >= 0 && N >= 0, PDM_E_BADARG, "%s: negative size", who);
    if (R == 0 || N == 0 || K == 0) return 0;
    PDM_REQUIRE(dZ && Yp && W && dX && dYout && coef && grads, PDM_E_BADARG, "%s: null pointer", who);
    PDM_REQUIRE(K % 8 == 0 && N % 8 == 0 && lddz % 8 == 0 && ldyp % 8 == 0 && ldw % 8 == 0 && lddx % 8 == 0 && lddy % 8 == 0 &&
                lddz >= K && ldyp >= K && ldw >= K && lddx >= N && lddy >= K, PDM_E_BADARG,
                "%s: K=%d N=%d and the strides must be multiples of 8 and cover the rows", who, K, N);
    PDM_REQUIRE(tg_al16(dZ) && tg_al16(Yp) && tg_al16(W) && tg_al16(dX) && tg_al16(dYout) && tg_al16(coef) && tg_al16(grads), PDM_E_BADARG,
                "%s: operands must be 16-byte aligned", who);
    PDM_REQUIRE(K <= TG_XFK, PDM_E_TOOLARGE, "%s: K=%d (<= %d)", who, K, TG_XFK);
    const bool bs = bstats != nullptr;
    if (bs) PDM_REQUIRE(Bx && bcoef && ldbx % 8 == 0 && ldbx >= N && tg_al16(Bx) && tg_al16(bcoef), PDM_E_BADARG,
                        "%s: Bx / bcoef missing or misaligned (ldbx=%lld)", who, ldbx);
    // tiles: 128 x 128 as pdm_tg_gemm_nt for wide outputs; 128 x 64 for N <= 64 (the 256-row tiles of the plain kernel would
    // hold 64 more registers of staged operands than a wave has left: two tensors travel per row here)
    const int bn = N <= 64 ? 64 : 128;
    const int ncol = (N + bn - 1) / bn;
    const int slots = tg_dy_slots(R, N);
    const unsigned wgs = (unsigned)slots * (unsigned)ncol;
    TgNtArgs a{};
    a.X = static_cast<const unsigned short *>(dZ); a.ldx = lddz; a.W = static_cast<const unsigned short *>(W); a.ldw = ldw;
    a.Y = static_cast<unsigned short *>(dX); a.ldy = lddx; a.xf = coef; a.gf = grads;
    a.Xa = static_cast<const unsigned short *>(Yp); a.ldxa = ldyp; a.Xo = static_cast<unsigned short *>(dYout); a.ldxo = lddy;
    a.Bx = static_cast<const unsigned short *>(Bx); a.ldbx = ldbx; a.bcf = bcoef; a.bstats = bstats;
    a.R = R; a.K = K; a.N = N;
    if (bn == 64) {
        if (bs) hipLaunchKernelGGL((tg_nt_kernel<2, 1, 2, true>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);
        else hipLaunchKernelGGL((tg_nt_kernel<2, 1, 2>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);
    } else {
        if (bs) hipLaunchKernelGGL((tg_nt_kernel<2, 2, 2, true>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);
        else hipLaunchKernelGGL((tg_nt_kernel<2, 2, 2>), dim3(wgs), dim3(TG_T), 0, as_stream(stream), a, slots);
    }
    return check_launch(who);
}

extern "C" int pdm_tg_gemm_nt_dy(void *stream, long long R, int K, int N, const void *dZ, long long lddz, const void *Yp, long long ldyp,
                                 const void *W, long long ldw, void *dX, long long lddx, void *dYout, long long lddy, const float *coef,
                                 const float *grads) {
    return tg_gemm_nt_dy_impl("tg_gemm_nt_dy", stream, R, K, N, dZ, lddz, Yp, ldyp, W, ldw, dX, lddx, dYout, lddy, coef, grads, nullptr, 0,
                              nullptr, nullptr);
}

// pdm_tg_gemm_nt_dy whose product dX is itself the gradient of relu(bn(Bx)) (an inner layer of a stack): the gradient statistics
// of THAT BatchNorm leave through the epilogue as in pdm_tg_gemm_nt_bs; bstats (pdm_tg_dy_stats_parts(R, N), N, 2).
extern "C" int pdm_tg_gemm_nt_dy_bs(void *stream, long long R, int K, int N, const void *dZ, long long lddz, const void *Yp, long long ldyp,
                                    const void *W, long long ldw, void *dX, long long lddx, void *dYout, long long lddy, const float *coef,
                                    const float *grads, const void *Bx, long long ldbx, const float *bcoef, float *bstats) {
    PDM_REQUIRE(bstats, PDM_E_BADARG, "tg_gemm_nt_dy_bs: null pointer");
    return tg_gemm_nt_dy_impl("tg_gemm_nt_dy_bs", stream, R, K, N, dZ, lddz, Yp, ldyp, W, ldw, dX, lddx, dYout, lddy, coef, grads, Bx, ldbx,
                              bcoef, bstats);
}

static long long tg_wgrad_slabs(long long R, int K, int N) {
    long long slabs = (R + 1023) / 1024;
    const long long cap = (24ll << 20) / ((long long)N * K * 4);
    if (slabs > cap) slabs = cap < 1 ? 1 : cap;
    if (slabs > 1024) slabs = 1024;                  // (what one fold launch sums per element: 16 groups x 64 parts)
    return slabs < 1 ? 1 : slabs;
}
extern "C" size_t pdm_tg_wgrad_ws_bytes(long long R, int K, int N) {
    if (R <= 0 || K <= 0 || N <= 0) return 0;
    const long long slabs = tg_wgrad_slabs(R, K, N);
    return ((size_t)slabs * N * K + tg_fold_scratch_floats(slabs, (long long)N * K)) * sizeof(float);
}

// dW (N, K) fp32 (+)= dY (R, N)^T bf16 . X (R, K) bf16, fp32 accumulation; workspace of pdm_tg_wgrad_ws_bytes(R, K, N) bytes.
extern "C" int pdm_tg_wgrad(void *stream, long long R, int K, int N, const void *dY, long long ldy, const void *X, long long ldx, float *dW,
                            int accumulate, void *workspace, size_t workspace_bytes, const float *x_bn_coef) {
    PDM_REQUIRE(R >= 0 && K >= 0 && N >= 0, PDM_E_BADARG, "tg_wgrad: negative size");
    if (K == 0 || N == 0) return 0;
    PDM_REQUIRE(dW, PDM_E_BADARG, "tg_wgrad: null pointer");
    if (R == 0) {
        if (!accumulate) {
            const hipError_t e = hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, as_stream(stream));
            PDM_REQUIRE(e == hipSuccess, PDM_E_BADARG, "tg_wgrad: memset failed");
        }
        return 0;
    }
    PDM_REQUIRE(dY && X && workspace, PDM_E_BADARG, "tg_wgrad: null pointer");
    PDM_REQUIRE(K % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= K && ldy >= N, PDM_E_BADARG,
                "tg_wgrad: K=%d N=%d ldx=%lld ldy=%lld must be multiples of 8 and cover the rows", K, N, ldx, ldy);
    PDM_REQUIRE(tg_al16(dY) && tg_al16(X) && tg_al16(workspace), PDM_E_BADARG, "tg_wgrad: operands must be 16-byte aligned");
    PDM_REQUIRE(workspace_bytes >= pdm_tg_wgrad_ws_bytes(R, K, N), PDM_E_BADARG, "tg_wgrad: workspace of %zu bytes, need %zu",
                workspace_bytes, pdm_tg_wgrad_ws_bytes(R, K, N));
    const long long slabs = tg_wgrad_slabs(R, K, N);
    long long rps = (R + slabs - 1) / slabs;
    rps = (rps + TG_WR - 1) / TG_WR * TG_WR;
    const long long used = (R + rps - 1) / rps;
    PDM_REQUIRE(used <= 65535, PDM_E_TOOLARGE, "tg_wgrad: %lld slabs", used);
    TgTnArgs a;
    a.dY = static_cast<const unsigned short *>(dY); a.ldy = ldy; a.X = static_cast<const unsigned short *>(X); a.ldx = ldx;
    a.partial = static_cast<float *>(workspace); a.xf = x_bn_coef; a.R = R; a.rows_per_slab = rps; a.N = N; a.K = K;
    const unsigned tiles = (unsigned)(((N + 127) / 128) * ((K + 127) / 128));
    if (N <= 64 && K <= 64) {       // narrow layers: the waves split the rows, one C x C result per workgroup
        const dim3 g(1, (unsigned)used);
        if (N <= 32 && K <= 32) {
            if (x_bn_coef) hipLaunchKernelGGL((tg_tn_narrow_kernel<32, true>), g, dim3(TG_T), 0, as_stream(stream), a);
            else hipLaunchKernelGGL((tg_tn_narrow_kernel<32, false>), g, dim3(TG_T), 0, as_stream(stream), a);
        } else {
            if (x_bn_coef) hipLaunchKernelGGL((tg_tn_narrow_kernel<64, true>), g, dim3(TG_T), 0, as_stream(stream), a);
            else hipLaunchKernelGGL((tg_tn_narrow_kernel<64, false>), g, dim3(TG_T), 0, as_stream(stream), a);
        }
    } else
    if (x_bn_coef) hipLaunchKernelGGL(tg_tn_kernel<true>, dim3(tiles * (unsigned)used), dim3(TG_T), 0, as_stream(stream), a);
    else hipLaunchKernelGGL(tg_tn_kernel<false>, dim3(tiles * (unsigned)used), dim3(TG_T), 0, as_stream(stream), a);
    int rc = check_launch("tg_wgrad");
    if (rc) return rc;
    const long long elems = (long long)N * K;
    return tg_fold(as_stream(stream), a.partial, (int)used, elems, dW, accumulate, a.partial + slabs * elems);
}

extern "C" size_t pdm_tg_colsum_ws_floats(long long R, int N) {
    if (R <= 0 || N <= 0) return 0;
    const long long slots = (R + 511) / 512 < 1024 ? (R + 511) / 512 : 1024;
    return (size_t)(slots * N) + tg_fold_scratch_floats(slots, N);
}

// out (N) fp32 = column sums of Y (R, N) bf16 (row stride ld, multiple of 8); N a multiple of 8, N <= 512; scratch of
// pdm_tg_colsum_ws_floats(R, N) floats.  Fixed summation order.
extern "C" int pdm_tg_colsum(void *stream, long long R, int N, const void *Y, long long ld, float *out, float *scratch) {
    PDM_REQUIRE(R >= 0 && N >= 0, PDM_E_BADARG, "tg_colsum: negative size");
    if (N == 0) return 0;
    PDM_REQUIRE(out, PDM_E_BADARG, "tg_colsum: null pointer");
    if (R == 0) {
        const hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, as_stream(stream));
        PDM_REQUIRE(e == hipSuccess, PDM_E_BADARG, "tg_colsum: memset failed");
        return 0;
    }
    PDM_REQUIRE(Y && scratch && N % 8 == 0 && N <= 512 && ld % 8 == 0 && ld >= N && tg_al16(Y), PDM_E_BADARG,
                "tg_colsum: N=%d (multiple of 8, <= 512), ld=%lld (multiple of 8), 16-byte aligned rows", N, ld);
    int chp = 1;
    while (chp * 8 < N) chp <<= 1;
    const long long slots = (R + 511) / 512 < 1024 ? (R + 511) / 512 : 1024;
    const long long rps = (R + slots - 1) / slots;
    hipLaunchKernelGGL(tg_colsum_kernel, dim3((unsigned)slots), dim3(256), 0, as_stream(stream), static_cast<const unsigned short *>(Y), ld, R, N,
                       chp, rps, scratch);
    int rc = check_launch("tg_colsum");
    if (rc) return rc;
    return tg_fold(as_stream(stream), scratch, (int)slots, N, out, 0, scratch + slots * N);
}

// W (N, K) fp32 -> bf16 copies: Wb (N, ldb) row-major and / or Wt (K, ldt) transposed (either may be null); pad columns zero.
extern "C" int pdm_tg_pack_weight(void *stream, int N, int K, const float *W, void *Wb, int ldb, void *Wt, int ldt) {
    PDM_REQUIRE(N >= 0 && K >= 0 && (!Wb || ldb >= K) && (!Wt || ldt >= N), PDM_E_BADARG, "tg_pack_weight: bad size");
    if (N == 0 || K == 0 || (!Wb && !Wt)) return 0;
    PDM_REQUIRE(W, PDM_E_BADARG, "tg_pack_weight: null pointer");
    long long elems = 0;
    if (Wb) elems = (long long)N * ldb;
    if (Wt && (long long)K * ldt > elems) elems = (long long)K * ldt;
    hipLaunchKernelGGL(tg_pack_weight_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, as_stream(stream), W, N, K,
                       static_cast<unsigned short *>(Wb), ldb, static_cast<unsigned short *>(Wt), ldt, N, K);
    return check_launch("tg_pack_weight");
}

// The pair a layer of the training path needs, in one launch and WHOLE: Wb (rows_to, cols_to) = W zero padded, Wt (cols_to, rows_to)
// = its transpose, every element written (no zero fill of the buffers in front).  rows_to >= N, cols_to >= K.
extern "C" int pdm_tg_pack_weight_pair(void *stream, int N, int K, const float *W, void *Wb, void *Wt, int rows_to, int cols_to) {
    PDM_REQUIRE(N >= 0 && K >= 0 && rows_to >= N && cols_to >= K, PDM_E_BADARG, "tg_pack_weight_pair: bad size");
    if (rows_to == 0 || cols_to == 0) return 0;
    PDM_REQUIRE(Wb && Wt && (W || N == 0 || K == 0), PDM_E_BADARG, "tg_pack_weight_pair: null pointer");
    const long long elems = (long long)rows_to * cols_to;
    hipLaunchKernelGGL(tg_pack_weight_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, as_stream(stream), W, N, K,
                       static_cast<unsigned short *>(Wb), cols_to, static_cast<unsigned short *>(Wt), rows_to, rows_to, cols_to);
    return check_launch("tg_pack_weight_pair");
}

// pdm_tg_pack_weight_pair for every layer of a model in ONE launch (a training step repacks ~40 parameters after each optimizer
// step: 40 launches of ~4.5 us).  jobs: `njobs` records in DEVICE memory, 48 bytes each, in this order:
//   { const float *W; void *Wb; void *Wt; int N, K, rows_to, cols_to; long long first_block; }
// first_block[0] = 0, first_block[j + 1] = first_block[j] + ceil(rows_to[j] cols_to[j] / 256); total_blocks = the sum.
extern "C" int pdm_tg_pack_weight_many(void *stream, int njobs, const void *jobs, long long total_blocks) {
    static_assert(sizeof(TgPackJob) == 48, "TgPackJob is part of the C ABI");
    PDM_REQUIRE(njobs >= 0 && total_blocks >= 0 && total_blocks <= 0x7fffffffll, PDM_E_BADARG, "tg_pack_weight_many: bad size");
    if (njobs == 0 || total_blocks == 0) return 0;
    PDM_REQUIRE(jobs, PDM_E_BADARG, "tg_pack_weight_many: null pointer");
    hipLaunchKernelGGL(tg_pack_weight_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream),
                       static_cast<const TgPackJob *>(jobs), njobs);
    return check_launch("tg_pack_weight_many");
}
