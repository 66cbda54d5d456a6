// Fused "gather -> shared MLP -> pool/store" kernels on fp32 MFMA for gfx950 (inference: BatchNorm
// folded into the 1x1-conv weights on the host).
//
// Replaces, for one SA scale, the chain of pointnet2_utils.py:250-257 (two grouping_operation calls,
// subtract, torch.cat) + pointnet2_modules.py:40-52 (3 x [Conv2d 1x1, BatchNorm2d, ReLU], max_pool2d)
// and, for one FP module, pointnet2_modules.py:158-170 (three_interpolate, cat, 2 x [conv, BN, ReLU]);
// the (B, C, M, nsample) tensors of the reference never exist in HBM.
//
// Design (CDNA4-first, not a CUDA tiling): ONE wave64 = one workgroup = one tile of 16 positions;
// waves are fully independent (no barriers).  The contraction is computed TRANSPOSED,
//     H_l^T [C_l x 16 positions] = W_l [C_l x K] * H_{l-1}^T [K x 16],
// with v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains):
//   A operand = weights, host-prepacked so lane (oc = l&15, g = l>>4) reads ONE float4 per 16-deep
//               K block = its A values of 4 consecutive MFMAs (1 KiB coalesced per wave instruction);
//   B operand = activations: lane (pos = l&15, g) needs H[pos][16kb + 4g + s], s = 0..3 = one
//               ds_read_b128 from the wave-private LDS tile H[pos][K] (or one 16-byte gather from a
//               point-major feature row in the first layer);
//   D         = lane (pos, g) holds output channels 16mb + 4g + i, i = 0..3 = exactly the float4 the
//               NEXT layer reads as its B operand, so the epilogue is bias + ReLU + one ds_write_b128.
// Features are point-major ((B, N, C), C contiguous) inside the fused path so a gathered neighbour is
// one contiguous row.  Max-pool over nsample = DPP row-max over the 16 lanes that share g.
#include "common.h"

#include <mutex>
#include <set>

namespace pdm {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int FM_MAXL = 4;

struct MlpDesc {
    int nlayers;
    int K[FM_MAXL + 1];  // padded (multiple of 16) channel counts: K[0] input, K[l] output of layer l
    int woff[FM_MAXL];   // float offsets of each layer's packed weights / bias
    int boff[FM_MAXL];
    int stage_in;        // 1: layer-1 input is gathered once into LDS; 0: re-gathered per output block pair
    int pool_floats;     // SA: width of the last layer (max-pool combine buffer), FP: 0
    int relu_last;       // 0: the last layer stores W x + b without the ReLU (pre-projection rows)
    int lds_p, lds_q;    // LDS tile widths (floats per position) of the two ping-pong buffers
    int swz;             // 1: rows are multiples of 64 floats and the 16-byte slots of a row are XOR-swizzled by the
                         //    position (lds_col); 0: round 1's layout, rows padded by 4 floats
};

// Column (in floats) of channel quad (blk, g) in the LDS row of position `pos`.  ds_read_b128 is served in four groups
// of 16 NON-contiguous lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... MI355X_MICROARCH.md, LDS) that must hit 16
// distinct 16-byte slots of the 256-byte bank row.  With rows padded by 4 floats (slot = pos + g mod 16) lanes 27 and 12
// of the first group meet on slot 12: every B-fragment read of the fused kernels was 2-way conflicted in part
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.44-0.47 in round 1's counters).  With slot = (4 blk + g) XOR pos inside
// each 256-byte segment the eight lanes with g even / odd of a group take {c ^ p} and {(c | 1) ^ p'} over position sets
// closed under p -> p ^ 1: sixteen distinct slots; ds_write_b128's groups of 8 contiguous lanes are distinct as well.
__device__ __forceinline__ int lds_col(int blk, int g, int pos, int swz) {
    const int q = 4 * blk + g;
    return swz ? (((q & ~15) | ((q ^ pos) & 15)) << 2) : (q << 2);
}

struct SaArgs {
    int b, n, m, cin, ns;
    const float *xyz;      // (B,N,3)
    const float *new_xyz;  // (B,M,3)
    const float *feat;     // (B,N,cin) point-major, may be null when cin == 0
    const int *idx;        // (B,M,ns)
    float *out;            // (B,M,out_stride) point-major
    int out_stride, out_coff, cout;
    const float *z;        // (B,N,z_stride) pre-projected layer-1 partial sums of the source points, or null
    int z_stride, z_coff;
};

struct FpArgs {
    int b, n, m, c_known, c_skip;
    const float *known;   // (B,m,c_known) point-major
    const float *skip;    // (B,n,c_skip) point-major, may be null
    const int *idx;       // (B,n,3)
    const float *weight;  // (B,n,3)
    float *out;           // (B,n,out_stride) point-major
    int out_stride, cout;
    const float *z;       // (B,m,z_stride) pre-projected layer-1 partial sums of the known points, or null
    int z_stride;
};

__device__ __forceinline__ f4 mfma4(f4 acc, f4 a, f4 b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

// max(v, floor) as one v_med3_f32 (fmaxf() costs a canonicalising v_max v,v,v on top; inline asm would hide the
// MFMA -> VALU read hazard from the compiler).  floor = 0 for ReLU, -inf for a linear last layer.
__device__ __forceinline__ float max1(float v, float floor) { return __builtin_amdgcn_fmed3f(v, floor, __builtin_inff()); }
__device__ __forceinline__ f4 floor4(f4 v, float floor) {
    v.x = max1(v.x, floor); v.y = max1(v.y, floor); v.z = max1(v.z, floor); v.w = max1(v.w, floor);
    return v;
}

// max over the 16 lanes of a DPP row (values are >= 0 after ReLU, so the integer image orders them)
__device__ __forceinline__ float row16_max_nonneg(float f) {
    int v = __builtin_bit_cast(int, f);
    asm volatile(
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __builtin_bit_cast(float, v);
}

// the same for four values at once: the four chains are interleaved so each DPP read of a just-written VGPR
// is three instructions behind its producer (the 2-wait-state hazard is covered without s_nops)
__device__ __forceinline__ f4 row16_max_nonneg4(f4 f) {
    // (copies first: __builtin_bit_cast applied to an ext-vector element lvalue reads element 0 for every lane)
    const float fx = f.x, fy = f.y, fz = f.z, fw = f.w;
    int a = __float_as_int(fx), b = __float_as_int(fy), c = __float_as_int(fz), d = __float_as_int(fw);
#define PDM_DPP4(CTRL)                                                        \
    "v_max_i32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
    "v_max_i32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
    "v_max_i32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
    "v_max_i32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" PDM_DPP4("quad_perm:[1,0,3,2]") PDM_DPP4("quad_perm:[2,3,0,1]")
                 PDM_DPP4("row_half_mirror") PDM_DPP4("row_mirror") "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef PDM_DPP4
    f.x = __int_as_float(a); f.y = __int_as_float(b); f.z = __int_as_float(c); f.w = __int_as_float(d);
    return f;
}

// max over aligned groups of L (1, 2, 4, 8, >= 16 -> 16) lanes of a DPP row = the first log2 L stages of the butterfly
// above; every lane of a group ends up with the group's maximum.  L is wave-uniform.
__device__ __forceinline__ f4 seg_max_nonneg4(f4 f, int L) {
    if (L <= 1) return f;
    const float fx = f.x, fy = f.y, fz = f.z, fw = f.w;
    int a = __float_as_int(fx), b = __float_as_int(fy), c = __float_as_int(fz), d = __float_as_int(fw);
#define PDM_DPP4(CTRL)                                                        \
    "v_max_i32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
    "v_max_i32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
    "v_max_i32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
    "v_max_i32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
    if (L >= 16)
        asm volatile("s_nop 1\n\t" PDM_DPP4("quad_perm:[1,0,3,2]") PDM_DPP4("quad_perm:[2,3,0,1]")
                     PDM_DPP4("row_half_mirror") PDM_DPP4("row_mirror") "s_nop 1"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else if (L == 8)
        asm volatile("s_nop 1\n\t" PDM_DPP4("quad_perm:[1,0,3,2]") PDM_DPP4("quad_perm:[2,3,0,1]")
                     PDM_DPP4("row_half_mirror") "s_nop 1"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else if (L == 4)
        asm volatile("s_nop 1\n\t" PDM_DPP4("quad_perm:[1,0,3,2]") PDM_DPP4("quad_perm:[2,3,0,1]") "s_nop 1"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else
        asm volatile("s_nop 1\n\t" PDM_DPP4("quad_perm:[1,0,3,2]") "s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef PDM_DPP4
    f.x = __int_as_float(a); f.y = __int_as_float(b); f.z = __int_as_float(c); f.w = __int_as_float(d);
    return f;
}

// ---- B-operand providers: float4 of channels [16kb + 4g, +4) of this lane's position ------------

struct LdsIn {
    const float *row;  // &H[pos][4g]
    __device__ __forceinline__ f4 operator()(int kb) const {
        return *reinterpret_cast<const f4 *>(row + 16 * kb);
    }
};

// SA: virtual row = [gathered feature row (cin), xyz[nb] - centre (3), zeros]
template <bool PRE>
struct SaIn;
template <>
struct SaIn<false> {
    const float *frow;  // feature row of this lane's neighbour (or null)
    float rx, ry, rz;
    int cin, g;
    bool vec;           // cin % 4 == 0 and 16-byte aligned rows
    struct Pre {};
    __device__ __forceinline__ float elem(int c) const {
        if (c < cin) return frow[c];
        const int e = c - cin;
        return e == 0 ? rx : e == 1 ? ry : e == 2 ? rz : 0.0f;
    }
    __device__ __forceinline__ f4 operator()(int kb) const {
        const int c0 = 16 * kb + 4 * g;
        if (vec && c0 + 4 <= cin) return *reinterpret_cast<const f4 *>(frow + c0);
        f4 v;
        v.x = elem(c0); v.y = elem(c0 + 1); v.z = elem(c0 + 2); v.w = elem(c0 + 3);
        return v;
    }
};
// hoisted form: virtual row = [xyz[nb] - centre (3), zeros]; the feature block of layer 1 arrives as z[nb]
template <>
struct SaIn<true> {
    const float *zrow;  // &z[b][nb][z_coff + 4g]
    float rx, ry, rz;   // zero on lanes with g != 0
    using Pre = f4;
    __device__ __forceinline__ f4 pre_load(int mb) const { return *reinterpret_cast<const f4 *>(zrow + 16 * mb); }
    __device__ __forceinline__ f4 pre_apply(const f4 &z, f4 acc) const { return acc + z; }
    __device__ __forceinline__ f4 operator()(int kb) const {
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (kb == 0) { v.x = rx; v.y = ry; v.z = rz; }
        return v;
    }
};

// FP: virtual row = [w0*known[i0] + w1*known[i1] + w2*known[i2] (c_known), skip row (c_skip), zeros];
// the interpolation keeps the oracle's rounding sequence fma(w2,p2, fma(w1,p1, rn(w0*p0))).
template <bool PRE>
struct FpIn;
template <>
struct FpIn<false> {
    const float *r0, *r1, *r2, *srow;
    float w0, w1, w2;
    int ck, cs, g;
    bool vec_k, vec_s, live;
    struct Pre {};
    __device__ __forceinline__ float interp(float a, float b, float c) const {
        return __fmaf_rn(w2, c, __fmaf_rn(w1, b, __fmul_rn(w0, a)));
    }
    __device__ __forceinline__ float elem(int c) const {
        if (c < ck) return interp(r0[c], r1[c], r2[c]);
        const int e = c - ck;
        return e < cs ? srow[e] : 0.0f;
    }
    __device__ __forceinline__ f4 operator()(int kb) const {
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (!live) return v;
        const int c0 = 16 * kb + 4 * g;
        if (vec_k && c0 + 4 <= ck) {
            const f4 a = *reinterpret_cast<const f4 *>(r0 + c0);
            const f4 b = *reinterpret_cast<const f4 *>(r1 + c0);
            const f4 c = *reinterpret_cast<const f4 *>(r2 + c0);
            v.x = interp(a.x, b.x, c.x); v.y = interp(a.y, b.y, c.y);
            v.z = interp(a.z, b.z, c.z); v.w = interp(a.w, b.w, c.w);
            return v;
        }
        if (vec_s && c0 >= ck && c0 + 4 <= ck + cs) return *reinterpret_cast<const f4 *>(srow + (c0 - ck));
        v.x = elem(c0); v.y = elem(c0 + 1); v.z = elem(c0 + 2); v.w = elem(c0 + 3);
        return v;
    }
};
// hoisted form: virtual row = [skip row (c_skip), zeros]; the known-feature block of layer 1 arrives as
// z rows of the three neighbours.  Interpolation is linear, so W (sum_k w_k f_k) = sum_k w_k (W f_k).
template <>
struct FpIn<true> {
    const float *zbase;   // &z[b][0][4g]
    const float *srow;
    int o0, o1, o2;       // float offsets of the three neighbours' z rows
    float w0, w1, w2;
    int cs, g;
    bool vec_s, live;
    struct Pre { f4 a, b, c; };
    __device__ __forceinline__ float interp(float a, float b, float c) const {
        return __fmaf_rn(w2, c, __fmaf_rn(w1, b, __fmul_rn(w0, a)));
    }
    __device__ __forceinline__ Pre pre_load(int mb) const {
        Pre p;
        p.a = p.b = p.c = f4{0.f, 0.f, 0.f, 0.f};
        if (live) {
            p.a = *reinterpret_cast<const f4 *>(zbase + o0 + 16 * mb);
            p.b = *reinterpret_cast<const f4 *>(zbase + o1 + 16 * mb);
            p.c = *reinterpret_cast<const f4 *>(zbase + o2 + 16 * mb);
        }
        return p;
    }
    __device__ __forceinline__ f4 pre_apply(const Pre &p, f4 acc) const {
        acc.x += interp(p.a.x, p.b.x, p.c.x); acc.y += interp(p.a.y, p.b.y, p.c.y);
        acc.z += interp(p.a.z, p.b.z, p.c.z); acc.w += interp(p.a.w, p.b.w, p.c.w);
        return acc;
    }
    __device__ __forceinline__ f4 operator()(int kb) const {
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (!live) return v;
        const int c0 = 16 * kb + 4 * g;
        if (vec_s && c0 + 4 <= cs) return *reinterpret_cast<const f4 *>(srow + c0);
        v.x = c0 < cs ? srow[c0] : 0.f; v.y = c0 + 1 < cs ? srow[c0 + 1] : 0.f;
        v.z = c0 + 2 < cs ? srow[c0 + 2] : 0.f; v.w = c0 + 3 < cs ? srow[c0 + 3] : 0.f;
        return v;
    }
};

// ---- sinks: receive the bias+ReLU'd float4 (channels 16mb + 4g .. +3 of this lane's position) ---

struct LdsOut {
    float *row;  // &H[pos][4g]
    __device__ __forceinline__ void operator()(int mb, f4 v) const {
        *reinterpret_cast<f4 *>(row + 16 * mb) = v;
    }
};

// SA epilogue: max over the 16 positions of the tile; tiles of one centre combine through `pool`.
struct PoolOut {
    float *pool;   // LDS, K_last floats, wave-private
    float *orow;   // &out[b][j][coff]
    int cout, lane, g;
    bool first_tile, last_tile;
    __device__ __forceinline__ void operator()(int mb, f4 v) const {
        v = row16_max_nonneg4(v);
        if ((lane & 15) != 0 || !orow) return;
        float *p = pool + 16 * mb + 4 * g;
        if (!first_tile) {
            const f4 o = *reinterpret_cast<const f4 *>(p);
            v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
        }
        if (!last_tile) {
            *reinterpret_cast<f4 *>(p) = v;
            return;
        }
        const int c0 = 16 * mb + 4 * g;
        if (c0 + 4 <= cout) {
            *reinterpret_cast<f4 *>(orow + c0) = v;  // host guarantees 16-byte alignment of orow
        } else {
            if (c0 < cout) orow[c0] = v.x;
            if (c0 + 1 < cout) orow[c0 + 1] = v.y;
            if (c0 + 2 < cout) orow[c0 + 2] = v.z;
        }
    }
};

// SA epilogue of the compacted form (sa_pack.hip): a tile of class L holds 16/L whole centres (L = 32: half of one);
// orow is set on the first lane of every live segment only.
struct SegPoolOut {
    float *pool;   // LDS, K_last floats: combines the tile pair of an L = 32 centre
    float *orow;   // &out[centre][coff] on writer lanes, else null
    int cout, g, L;
    bool first_tile, last_tile;
    __device__ __forceinline__ void operator()(int mb, f4 v) const {
        v = seg_max_nonneg4(v, L);
        if (!orow) return;
        float *p = pool + 16 * mb + 4 * g;
        if (!first_tile) {
            const f4 o = *reinterpret_cast<const f4 *>(p);
            v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
        }
        if (!last_tile) {
            *reinterpret_cast<f4 *>(p) = v;
            return;
        }
        const int c0 = 16 * mb + 4 * g;
        if (c0 + 4 <= cout) {
            *reinterpret_cast<f4 *>(orow + c0) = v;
        } else {
            if (c0 < cout) orow[c0] = v.x;
            if (c0 + 1 < cout) orow[c0 + 1] = v.y;
            if (c0 + 2 < cout) orow[c0 + 2] = v.z;
        }
    }
};

struct RowOut {
    float *orow;  // &out[b][p][0], null for positions past n
    int cout, g;
    __device__ __forceinline__ void operator()(int mb, f4 v) const {
        if (!orow) return;
        const int c0 = 16 * mb + 4 * g;
        if (c0 + 4 <= cout) {
            *reinterpret_cast<f4 *>(orow + c0) = v;
        } else {
            if (c0 < cout) orow[c0] = v.x;
            if (c0 + 1 < cout) orow[c0 + 1] = v.y;
            if (c0 + 2 < cout) orow[c0 + 2] = v.z;
        }
    }
};

// A workgroup iteration covers NT (1 or 2) tiles of 16 positions that share every A (weight) fragment.
template <class T, int NT>
struct Tiles {
    T t[NT];
};
template <int NT>
struct LdsTilesIn {
    struct Pre {};
    const float *row;  // &H[pos][0] of tile 0; tile t is 16 positions further
    int tile_stride;   // floats between tiles = 16 * width
    int g, pos, swz;
    __device__ __forceinline__ f4 operator()(int t, int kb) const {
        return *reinterpret_cast<const f4 *>(row + t * tile_stride + lds_col(kb, g, pos, swz));
    }
};
template <int NT>
struct LdsTilesOut {
    float *row;
    int tile_stride;
    int g, pos, swz;
    __device__ __forceinline__ void operator()(int t, int mb, f4 v) const {
        *reinterpret_cast<f4 *>(row + t * tile_stride + lds_col(mb, g, pos, swz)) = v;
    }
};
template <class T, int NT>
struct TilesIn {
    using Pre = typename T::Pre;
    const Tiles<T, NT> *p;
    __device__ __forceinline__ f4 operator()(int t, int kb) const { return p->t[t](kb); }
    __device__ __forceinline__ Pre pre_load(int t, int mb) const { return p->t[t].pre_load(mb); }
    __device__ __forceinline__ f4 pre_apply(int t, const Pre &z, f4 acc) const { return p->t[t].pre_apply(z, acc); }
};
// layer-1 input staged in LDS, pre-projection still taken from the gather provider
template <class T, int NT>
struct StagedPreIn {
    using Pre = typename T::Pre;
    LdsTilesIn<NT> li;
    const Tiles<T, NT> *p;
    __device__ __forceinline__ f4 operator()(int t, int kb) const { return li(t, kb); }
    __device__ __forceinline__ Pre pre_load(int t, int mb) const { return p->t[t].pre_load(mb); }
    __device__ __forceinline__ f4 pre_apply(int t, const Pre &z, f4 acc) const { return p->t[t].pre_apply(z, acc); }
};
template <class T, int NT>
struct TilesOut {
    const Tiles<T, NT> *p;
    __device__ __forceinline__ void operator()(int t, int mb, f4 v) const { p->t[t](mb, v); }
};

// One pass = NB (1..4) 16-channel output blocks starting at block mb, over the whole K range, for NT
// position tiles.  Per 16-deep K block: NT B fragments + NB A fragments feed 4*NB*NT MFMAs, issued
// round-robin over the NB*NT independent accumulators (a dependent v_mfma_f32_16x16x4_f32 needs 40
// cycles, the pipe issues one per 32); the next K block's fragments are requested before the current
// block's MFMAs.
template <int NB, int NT, bool PRE, bool PP, class In, class Out>
__device__ __forceinline__ void mlp_pass(int nkb, int mb, const f4 *__restrict__ w, size_t bstride,
                                         const float *__restrict__ bias, int g, bool relu, const In &in,
                                         const Out &out) {
    const f4 *__restrict__ wb[NB];
    f4 acc[NB][NT], a[NB], an[NB], b[NT], bn[NT];
    const float floor = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(relu ? 0 : (int)0xff800000u));
    // pre-projected part of the first layer: requested first, consumed after the K loop
    typename In::Pre z[PRE ? NB : 1][PRE ? NT : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) z[i][t] = in.pre_load(t, mb + i);
    }
    if constexpr (!PP) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        wb[i] = w + (size_t)(mb + i) * bstride;
        const f4 bi = *reinterpret_cast<const f4 *>(bias + 16 * (mb + i) + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = bi;
#if defined(FUSED_DIAG) && (FUSED_DIAG == 1 || FUSED_DIAG == 4)
        wb[i] = w;  // timing-only: every pass re-reads the same L1-resident weight fragments
#endif
        an[i] = wb[i][0];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) bn[t] = in(t, 0);
    for (int kb = 0; kb < nkb; ++kb) {
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = bn[t];
#pragma unroll
        for (int i = 0; i < NB; ++i) a[i] = an[i];
        if (kb + 1 < nkb) {
#if defined(FUSED_DIAG) && (FUSED_DIAG == 1 || FUSED_DIAG == 4)
            const size_t o = 0;
#else
            const size_t o = (size_t)(kb + 1) * 64;
#endif
#if !(defined(FUSED_DIAG) && (FUSED_DIAG == 2 || FUSED_DIAG == 4))
#pragma unroll
            for (int t = 0; t < NT; ++t) bn[t] = in(t, kb + 1);
#endif
#pragma unroll
            for (int i = 0; i < NB; ++i) an[i] = wb[i][o];
        }
#if defined(FUSED_DIAG) && FUSED_DIAG == 3
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[i][t] += a[i] * b[t];  // timing-only: no MFMA
        continue;
#endif
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[t].x, acc[i][t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[t].y, acc[i][t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[t].z, acc[i][t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[t].w, acc[i][t], 0, 0, 0);
    }
    } else {
    // K loop unrolled by two over two fixed fragment sets (no register rotation): the set a k-block has just
    // consumed is refilled for k-block + 2 right behind its MFMAs, so every fragment load has the MFMAs of two
    // k-blocks (8 * NB * NT of them) to land in.
    f4 a1[NB], b1[NT];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        wb[i] = w + (size_t)(mb + i) * bstride;
        const f4 bi = *reinterpret_cast<const f4 *>(bias + 16 * (mb + i) + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = bi;
        a[i] = wb[i][0];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = in(t, 0);
    if (nkb > 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t) b1[t] = in(t, 1);
#pragma unroll
        for (int i = 0; i < NB; ++i) a1[i] = wb[i][64];
    }
#define PDM_MFMA_BLOCK(A, B)                                                                              \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) _Pragma("unroll") for (int t = 0; t < NT; ++t)          \
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i].x, B[t].x, acc[i][t], 0, 0, 0);              \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) _Pragma("unroll") for (int t = 0; t < NT; ++t)          \
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i].y, B[t].y, acc[i][t], 0, 0, 0);              \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) _Pragma("unroll") for (int t = 0; t < NT; ++t)          \
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i].z, B[t].z, acc[i][t], 0, 0, 0);              \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) _Pragma("unroll") for (int t = 0; t < NT; ++t)          \
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i].w, B[t].w, acc[i][t], 0, 0, 0);
    int kb = 0;
    for (; kb + 2 <= nkb; kb += 2) {
        PDM_MFMA_BLOCK(a, b)
        if (kb + 2 < nkb) {
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = in(t, kb + 2);
#pragma unroll
            for (int i = 0; i < NB; ++i) a[i] = wb[i][(size_t)(kb + 2) * 64];
        }
        PDM_MFMA_BLOCK(a1, b1)
        if (kb + 3 < nkb) {
#pragma unroll
            for (int t = 0; t < NT; ++t) b1[t] = in(t, kb + 3);
#pragma unroll
            for (int i = 0; i < NB; ++i) a1[i] = wb[i][(size_t)(kb + 3) * 64];
        }
    }
    if (kb < nkb) { PDM_MFMA_BLOCK(a, b) }
#undef PDM_MFMA_BLOCK
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f4 v = acc[i][t];
            if constexpr (PRE) v = in.pre_apply(t, z[i][t], v);
            out(t, mb + i, floor4(v, floor));
        }
}

// One layer for the W waves of a workgroup: the nmb output blocks are dealt to the waves in contiguous
// shares; a wave walks its share in passes of up to 4 blocks.  `wave` must be wave-uniform (SGPR).
template <int W, int NT, int MAXNB, bool PRE, class In, class Out>
__device__ __forceinline__ void mlp_layer(int nkb, int nmb, const float *__restrict__ wp,
                                          const float *__restrict__ bias, int lane, int wave, bool relu,
                                          const In &in, const Out &out) {
    const int g = lane >> 4;
    const int share = (nmb + W - 1) / W;                  // blocks per wave
    const int mb_begin = wave * share;
    const int mb_end = min(mb_begin + share, nmb);
    if (mb_begin >= mb_end) return;
    const int passes = (share + MAXNB - 1) / MAXNB;
    const int bpp = (share + passes - 1) / passes;        // blocks per pass, 1..MAXNB
    const f4 *__restrict__ w = reinterpret_cast<const f4 *>(wp) + lane;
    const size_t bstride = (size_t)nkb * 64;              // f4 elements between consecutive blocks
#if defined(FUSED_DIAG)
    constexpr bool PP = false;
#else
    constexpr bool PP = W >= 4;  // wide layers: two-set K loop (measured: SA3 -10 %, SA4 -3 %; narrow layers +3 %)
#endif
    for (int mb = mb_begin; mb < mb_end; mb += bpp) {
        const int nb = min(bpp, mb_end - mb);             // wave-uniform
        if (MAXNB >= 4 && nb == 4) mlp_pass<4, NT, PRE, PP>(nkb, mb, w, bstride, bias, g, relu, in, out);
        else if (MAXNB >= 3 && nb == 3) mlp_pass<3, NT, PRE, PP>(nkb, mb, w, bstride, bias, g, relu, in, out);
        else if (MAXNB >= 2 && nb == 2) mlp_pass<2, NT, PRE, PP>(nkb, mb, w, bstride, bias, g, relu, in, out);
        else mlp_pass<1, NT, PRE, PP>(nkb, mb, w, bstride, bias, g, relu, in, out);
    }
}

template <int W>
__device__ __forceinline__ void wg_sync() {
    if (W > 1) __syncthreads();
    else __builtin_amdgcn_wave_barrier();  // one wave: LDS ops execute in order, only pin the compiler
}

// Runs layer L for the NT tiles of one workgroup iteration.  P/Q = the workgroup's two LDS buffers
// (NT*16 positions x lds_p / lds_q floats): layer 1 writes Q, layer 2 writes P, layer 3 writes Q ...; a
// staged input lives in P.  The layer sequence is unrolled so every descriptor field is read with a
// constant index (a runtime index into the by-value descriptor would push it to scratch).
template <int W, int NT, int MAXNB, int PSW, int L, bool PRE, class T, class Out>
__device__ __forceinline__ void run_layer(const MlpDesc &d, const float *__restrict__ wpack,
                                          const float *__restrict__ bias, float *P, float *Q, int lane,
                                          int wave, const TilesIn<T, NT> &in, const Out &out) {
    const int pos = lane & 15, g = lane >> 4;
    const int nkb = d.K[L - 1] >> 4, nmb = d.K[L] >> 4;
    const float *wl = wpack + d.woff[L - 1];
    const float *bl = bias + d.boff[L - 1];
    const bool last = L == d.nlayers;
    const bool relu = !last || d.relu_last;
    float *ob = (L & 1) ? Q : P;
    const int ow = (L & 1) ? d.lds_q : d.lds_p;
    const LdsTilesOut<NT> lo{ob + pos * ow, 16 * ow, g, pos, d.swz};
    if (L == 1 && !d.stage_in) {
        if (last) mlp_layer<W, NT, MAXNB, PRE>(nkb, nmb, wl, bl, lane, wave, relu, in, out);
        else mlp_layer<W, NT, MAXNB, PRE>(nkb, nmb, wl, bl, lane, wave, relu, in, lo);
    } else {
        float *ib = (L & 1) ? P : Q;  // layer 1 (staged) and layer 3 read P, layer 2 reads Q
        const int iw = (L & 1) ? d.lds_p : d.lds_q;
        const LdsTilesIn<NT> li{ib + pos * iw, 16 * iw, g, pos, d.swz};
        if constexpr (L == 1 && PRE) {
            const StagedPreIn<T, NT> si{li, in.p};
            if (last) mlp_layer<W, NT, MAXNB, true>(nkb, nmb, wl, bl, lane, wave, relu, si, out);
            else mlp_layer<W, NT, MAXNB, true>(nkb, nmb, wl, bl, lane, wave, relu, si, lo);
        } else {
            if (last) mlp_layer<W, NT, MAXNB, false>(nkb, nmb, wl, bl, lane, wave, relu, li, out);
            else mlp_layer<W, NT, MAXNB, false>(nkb, nmb, wl, bl, lane, wave, relu, li, lo);
        }
    }
    wg_sync<W * PSW>();
}

template <int W, int NT, int MAXNB, int PSW, bool PRE, class T, class Out>
__device__ __forceinline__ void run_mlp(const MlpDesc &d, const float *__restrict__ wpack,
                                        const float *__restrict__ bias, float *P, float *Q, int lane,
                                        int wave, const TilesIn<T, NT> &in, const Out &out) {
    if (d.stage_in) {
        // gather the input tiles once, K blocks dealt round-robin to the waves: P[tile*16 + pos][K0]
        float *row = P + (lane & 15) * d.lds_p;
        const int nkb0 = d.K[0] >> 4;
        for (int kb = wave; kb < nkb0; kb += W)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                *reinterpret_cast<f4 *>(row + t * 16 * d.lds_p + lds_col(kb, lane >> 4, lane & 15, d.swz)) = in(t, kb);
        wg_sync<W * PSW>();
    }
    run_layer<W, NT, MAXNB, PSW, 1, PRE>(d, wpack, bias, P, Q, lane, wave, in, out);
    if (d.nlayers >= 2) run_layer<W, NT, MAXNB, PSW, 2, false>(d, wpack, bias, P, Q, lane, wave, in, out);
    if (d.nlayers >= 3) run_layer<W, NT, MAXNB, PSW, 3, false>(d, wpack, bias, P, Q, lane, wave, in, out);
    if (d.nlayers >= 4) run_layer<W, NT, MAXNB, PSW, 4, false>(d, wpack, bias, P, Q, lane, wave, in, out);
}

// Workgroup = PSW position groups x W channel-split waves.  The W waves of a group share one LDS tile
// set and split each layer's output blocks; the PSW groups work on different tiles but walk the same
// (layer, block, K) sequence, so the weight fragments one group pulls from L2 are L1 hits for the others
// (the workgroup-wide barrier after every layer keeps them together).
template <int W, int NT, int MAXNB, int PSW, bool PRE>
__global__ __launch_bounds__(64 * W * PSW) void sa_mlp_fused_kernel(MlpDesc d, SaArgs a,
                                                                    const float *__restrict__ wpack,
                                                                    const float *__restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave_all % W, grp = wave_all / W;
    const int pos = lane & 15, g = lane >> 4;
    const int region = NT * 16 * (d.lds_p + d.lds_q) + d.pool_floats;
    float *P = lds + grp * region;
    float *Q = P + NT * 16 * d.lds_p;
    float *pool = Q + NT * 16 * d.lds_q;
    const int tpc = a.ns >> 4;  // tiles per centre
    const long long ncentres = (long long)a.b * a.m;
    // Work unit = whole centres, so the tiles of one centre (pooled through `pool`) stay in one workgroup:
    // tpc >= NT: one centre per unit, walked in ceil(tpc/NT) sub-steps; tpc < NT: NT centres per unit.
    const int cpu_ = tpc >= NT ? 1 : NT / tpc;          // centres per unit
    const int nsub = tpc >= NT ? (tpc + NT - 1) / NT : 1;
    const long long nunits = (ncentres + cpu_ - 1) / cpu_;
    for (long long base = (long long)blockIdx.x * PSW; base < nunits; base += (long long)gridDim.x * PSW) {
        const long long unit = base + grp;  // groups past the end run dead tiles so barriers still match
        for (int sub = 0; sub < nsub; ++sub) {
            Tiles<SaIn<PRE>, NT> in;
            Tiles<PoolOut, NT> out;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const long long ctr_raw = tpc >= NT ? unit : unit * cpu_ + t / tpc;
                const int tic = tpc >= NT ? sub * NT + t : t % tpc;
                const bool live = unit < nunits && ctr_raw < ncentres && tic < tpc;
                const long long ctr = live ? ctr_raw : 0;
                const int tl = live ? tic : 0;
                const int b = (int)(ctr / a.m);
                const float *c3 = a.new_xyz + ctr * 3;
                const int nb = a.idx[ctr * a.ns + tl * 16 + pos];
                const float *p3 = a.xyz + ((size_t)b * a.n + nb) * 3;
                const float rx = p3[0] - c3[0], ry = p3[1] - c3[1], rz = p3[2] - c3[2];  // pointnet2_utils.py:252
                if constexpr (PRE) {
                    in.t[t].zrow = a.z + ((size_t)b * a.n + nb) * a.z_stride + a.z_coff + 4 * g;
                    in.t[t].rx = g == 0 ? rx : 0.f; in.t[t].ry = g == 0 ? ry : 0.f; in.t[t].rz = g == 0 ? rz : 0.f;
                } else {
                    in.t[t].frow = a.cin > 0 ? a.feat + ((size_t)b * a.n + nb) * a.cin : nullptr;
                    in.t[t].rx = rx; in.t[t].ry = ry; in.t[t].rz = rz;
                    in.t[t].cin = a.cin; in.t[t].g = g;
                    in.t[t].vec = (a.cin & 3) == 0 && a.cin > 0;
                }
                out.t[t].pool = pool;
                out.t[t].orow = live ? a.out + ctr * a.out_stride + a.out_coff : nullptr;
                out.t[t].cout = a.cout; out.t[t].lane = lane; out.t[t].g = g;
                out.t[t].first_tile = tic == 0; out.t[t].last_tile = tic == tpc - 1;
            }
            run_mlp<W, NT, MAXNB, PSW, PRE>(d, wpack, bias, P, Q, lane, wave, TilesIn<SaIn<PRE>, NT>{&in}, TilesOut<PoolOut, NT>{&out});
        }
    }
}

// class (segment length) of the tile that starts at `row`: classes are stored in ascending L, meta[k] = first row of class k
__device__ __forceinline__ int pack_tile_L(int row, int m1, int m2, int m3, int m4, int m5) {
    return 1 << ((row >= m1) + (row >= m2) + (row >= m3) + (row >= m4) + (row >= m5));
}

// The same SA scale over the compacted row list of sa_pack.hip: work unit = an aligned PAIR of 16-row tiles (the two
// tiles of an L = 32 centre pool through `pool`; every other tile is self-contained), NT = 2 takes the pair at once,
// NT = 1 in two sub-steps.  The number of pairs is read from meta[6] (device memory): the grid is sized for the worst
// case and surplus workgroups leave at once.
template <int W, int NT, int MAXNB, int PSW, bool PRE>
__device__ __forceinline__ void sa_packed_body(const MlpDesc &d, const SaArgs &a, const int2 *__restrict__ pack,
                                               const int *__restrict__ meta, const float *__restrict__ wpack,
                                               const float *__restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave_all % W, grp = wave_all / W;
    const int pos = lane & 15, g = lane >> 4;
    const int region = NT * 16 * (d.lds_p + d.lds_q) + d.pool_floats;
    float *P = lds + grp * region;
    float *Q = P + NT * 16 * d.lds_p;
    float *pool = Q + NT * 16 * d.lds_q;
    const int m1 = meta[1], m2 = meta[2], m3 = meta[3], m4 = meta[4], m5 = meta[5];
    const int npairs = meta[6] >> 5;
    constexpr int NSUB = 2 / NT;
    for (int base = blockIdx.x * PSW; base < npairs; base += gridDim.x * PSW) {
        const int unit = base + grp;  // groups past the end run dead tiles so barriers still match
        for (int sub = 0; sub < NSUB; ++sub) {
            Tiles<SaIn<PRE>, NT> in;
            Tiles<SegPoolOut, NT> out;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tile = unit * 2 + sub * NT + t;
                const bool live = unit < npairs;
                const int2 e = pack[live ? tile * 16 + pos : 0];
                const int L = pack_tile_L(tile * 16, m1, m2, m3, m4, m5);
                const int ctr = live ? e.y : -1;
                const float *c3 = a.new_xyz + (size_t)(ctr >= 0 ? ctr : 0) * 3;
                const float *p3 = a.xyz + (size_t)e.x * 3;
                const float rx = p3[0] - c3[0], ry = p3[1] - c3[1], rz = p3[2] - c3[2];  // pointnet2_utils.py:252
                if constexpr (PRE) {
                    in.t[t].zrow = a.z + (size_t)e.x * a.z_stride + a.z_coff + 4 * g;
                    in.t[t].rx = g == 0 ? rx : 0.f; in.t[t].ry = g == 0 ? ry : 0.f; in.t[t].rz = g == 0 ? rz : 0.f;
                } else {
                    in.t[t].frow = a.cin > 0 ? a.feat + (size_t)e.x * a.cin : nullptr;
                    in.t[t].rx = rx; in.t[t].ry = ry; in.t[t].rz = rz;
                    in.t[t].cin = a.cin; in.t[t].g = g;
                    in.t[t].vec = (a.cin & 3) == 0 && a.cin > 0;
                }
                const bool writer = ctr >= 0 && (pos & ((L < 16 ? L : 16) - 1)) == 0;
                out.t[t].pool = pool;
                out.t[t].orow = writer ? a.out + (size_t)ctr * a.out_stride + a.out_coff : nullptr;
                out.t[t].cout = a.cout; out.t[t].g = g; out.t[t].L = L;
                out.t[t].first_tile = L < 32 || (tile & 1) == 0;
                out.t[t].last_tile = L < 32 || (tile & 1) == 1;
            }
            run_mlp<W, NT, MAXNB, PSW, PRE>(d, wpack, bias, P, Q, lane, wave, TilesIn<SaIn<PRE>, NT>{&in}, TilesOut<SegPoolOut, NT>{&out});
        }
    }
}
template <int W, int NT, int MAXNB, int PSW, bool PRE>
__global__ __launch_bounds__(64 * W * PSW) void sa_packed_fused_kernel(MlpDesc d, SaArgs a,
                                                                       const int2 *__restrict__ pack,
                                                                       const int *__restrict__ meta,
                                                                       const float *__restrict__ wpack,
                                                                       const float *__restrict__ bias) {
    sa_packed_body<W, NT, MAXNB, PSW, PRE>(d, a, pack, meta, wpack, bias);
}
// The two scales of an SA level in ONE launch (blockIdx.y = scale): with compacted lists a deep level holds a few hundred row
// tiles per scale (SA4 on BASELINE's uniform clouds: 64 tile pairs = 64 busy workgroups on 256 CUs, each a ~65 us chain of
// weight fragments), so two launches one after the other leave most of the chip idle twice; side by side they take the time of
// one.  Both scales must map to the same instantiation (same waves / tiles / groups / hoisting); results are bit-identical.
struct SaScale {
    MlpDesc d;
    SaArgs a;
    const int2 *pack;
    const int *meta;
    const float *wpack, *bias;
};
template <int W, int NT, int MAXNB, int PSW, bool PRE>
__global__ __launch_bounds__(64 * W * PSW) void sa_packed_pair_kernel(SaScale s0, SaScale s1) {
    if (blockIdx.y == 0) sa_packed_body<W, NT, MAXNB, PSW, PRE>(s0.d, s0.a, s0.pack, s0.meta, s0.wpack, s0.bias);
    else sa_packed_body<W, NT, MAXNB, PSW, PRE>(s1.d, s1.a, s1.pack, s1.meta, s1.wpack, s1.bias);
}

template <int W, int NT, int MAXNB, int PSW, bool PRE>
__global__ __launch_bounds__(64 * W * PSW) void fp_mlp_fused_kernel(MlpDesc d, FpArgs a,
                                                                    const float *__restrict__ wpack,
                                                                    const float *__restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave_all % W, grp = wave_all / W;
    const int pos = lane & 15, g = lane >> 4;
    float *P = lds + grp * (NT * 16 * (d.lds_p + d.lds_q));
    float *Q = P + NT * 16 * d.lds_p;
    const int tps = (a.n + 15) >> 4;  // tiles per sample
    const long long ntiles = (long long)a.b * tps;
    const long long niter = (ntiles + NT - 1) / NT;
    // hoisted form: every fine point gathers z rows of ITS cloud's known set.  Workgroups go to the 8 XCDs round-robin
    // by linear id; when the iterations divide evenly, XCD x works through clouds x, x+8, ... so that its L2 holds the
    // known sets in flight instead of seeing every cloud's (same permutation as rows_gemm.hip)
    const int ipc = tps / NT;   // iterations per cloud
    const bool xcd_order = PRE && PSW == 1 && ipc * NT == tps && (a.b & 7) == 0 && (gridDim.x & 7) == 0 && a.b >= 8;
    for (long long base = (long long)blockIdx.x * PSW; base < niter; base += (long long)gridDim.x * PSW) {
        long long it = base + grp;
        if (xcd_order && it < niter) {
            const long long slot = it >> 3;
            it = ((slot / ipc) * 8 + (it & 7)) * ipc + slot % ipc;
        }
        Tiles<FpIn<PRE>, NT> in;
        Tiles<RowOut, NT> out;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long long tile = it < niter ? it * NT + t : ntiles;
            const int b = tile < ntiles ? (int)(tile / tps) : 0;
            const int p = (int)(tile - (long long)b * tps) * 16 + pos;
            FpIn<PRE> &f = in.t[t];
            f.live = tile < ntiles && p < a.n;
            const size_t q = (size_t)b * a.n + (f.live ? p : 0);
            f.w0 = f.w1 = f.w2 = 0.0f;
            f.srow = a.c_skip > 0 ? a.skip + q * a.c_skip : nullptr;
            f.cs = a.c_skip; f.g = g;
            if constexpr (PRE) {
                const int *id = a.idx + q * 3;
                const float *w = a.weight + q * 3;
                f.zbase = a.z + (size_t)b * a.m * a.z_stride + 4 * g;
                f.o0 = id[0] * a.z_stride; f.o1 = id[1] * a.z_stride; f.o2 = id[2] * a.z_stride;
                f.w0 = w[0]; f.w1 = w[1]; f.w2 = w[2];
                f.vec_s = (a.c_skip & 3) == 0 && a.c_skip > 0;
            } else {
                f.r0 = f.r1 = f.r2 = nullptr;
                if (a.idx) {  // null for plain rows (pdm_rows_mlp_fused)
                    const int *id = a.idx + q * 3;
                    const float *w = a.weight + q * 3;
                    f.r0 = a.known + ((size_t)b * a.m + id[0]) * a.c_known;
                    f.r1 = a.known + ((size_t)b * a.m + id[1]) * a.c_known;
                    f.r2 = a.known + ((size_t)b * a.m + id[2]) * a.c_known;
                    f.w0 = w[0]; f.w1 = w[1]; f.w2 = w[2];
                }
                f.ck = a.c_known;
                f.vec_k = (a.c_known & 3) == 0;
                f.vec_s = (a.c_skip & 3) == 0 && (a.c_known & 3) == 0 && a.c_skip > 0;
            }
            out.t[t].orow = f.live ? a.out + q * a.out_stride : nullptr;
            out.t[t].cout = a.cout; out.t[t].g = g;
        }
        run_mlp<W, NT, MAXNB, PSW, PRE>(d, wpack, bias, P, Q, lane, wave, TilesIn<FpIn<PRE>, NT>{&in}, TilesOut<RowOut, NT>{&out});
    }
}


// Register-resident form for small SA scales (input <= 16 channels, three layers of at most 64 outputs — SA1 of
// PointNet2MSG): one wave = one pair of 16-position tiles, ALL weight fragments live in VGPRs for the whole kernel
// (B1 + B2*B1 + B3*B2 float4 per lane), and because a wave owns every output block of its tiles the bias+ReLU'd
// accumulator of block mb IS the next layer's B fragment of k-block mb: no LDS, no barriers, no weight traffic
// after the first instruction — per tile only the index/neighbour gather, the MFMA chain and the DPP max-pool.
template <int B1, int B2, int B3, bool SAME>   // SAME: nsample >= 32, the two tiles of a step share their centre
__global__ __launch_bounds__(256) void sa_reg_mlp_kernel(SaArgs a, const float *__restrict__ wpack,
                                                         const float *__restrict__ bias) {
    constexpr int NT = 2;
    const int lane = threadIdx.x & 63, pos = lane & 15, g = lane >> 4;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const f4 *__restrict__ w1 = reinterpret_cast<const f4 *>(wpack) + lane;
    const f4 *__restrict__ w2 = w1 + 64 * B1;            // layer 1: B1 blocks x 1 k-block x 64 lanes
    const f4 *__restrict__ w3 = w2 + 64 * B2 * B1;
    f4 A1[B1], A2[B2][B1], A3[B3][B2];
#pragma unroll
    for (int mb = 0; mb < B1; ++mb) A1[mb] = w1[64 * mb];
#pragma unroll
    for (int mb = 0; mb < B2; ++mb)
#pragma unroll
        for (int kb = 0; kb < B1; ++kb) A2[mb][kb] = w2[64 * (mb * B1 + kb)];
#pragma unroll
    for (int mb = 0; mb < B3; ++mb)
#pragma unroll
        for (int kb = 0; kb < B2; ++kb) A3[mb][kb] = w3[64 * (mb * B2 + kb)];
    const float *__restrict__ bias1 = bias + 4 * g, *__restrict__ bias2 = bias1 + 16 * B1, *__restrict__ bias3 = bias2 + 16 * B2;
    const int tpc = a.ns >> 4;
    const int ncentres = a.b * a.m;
    const int cpu_ = SAME ? 1 : NT / tpc;               // centres per unit
    const int nsub = SAME ? (tpc + NT - 1) / NT : 1;
    const int nunits = (ncentres + cpu_ - 1) / cpu_;
#define PDM_CHAIN(ACC, A, B)                                                          \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).x, (B).x, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).y, (B).y, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).z, (B).z, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).w, (B).w, ACC, 0, 0, 0);
    // (unit, sub-step) sequence of this wave in 32-bit arithmetic (the host checks b*m*nsample < 2^31); the
    // neighbour indices of the next step are requested before the current step's MFMA chain (deeper prefetching
    // costs registers, i.e. waves per SIMD: measured slower).
    struct Where { int ctr, tic; bool live; };
    auto where = [&](int unit, int sub, int t) {
        const int ctr_raw = SAME ? unit : unit * cpu_ + t / tpc;
        const int tic = SAME ? sub * NT + t : t % tpc;
        Where r;
        r.live = unit < nunits && ctr_raw < ncentres && tic < tpc;
        r.ctr = r.live ? ctr_raw : 0;
        r.tic = r.live ? tic : 0;
        return r;
    };
    int nb_next[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const Where wq = where(wave_id, 0, t);
        nb_next[t] = a.idx[wq.ctr * a.ns + wq.tic * 16 + pos];
    }
    f4 best[SAME ? 1 : NT][B3];
    for (int unit = wave_id; unit < nunits; unit += nwaves)
    for (int sub = 0; sub < nsub; ++sub) {
        if (sub == 0) {
#pragma unroll
            for (int t = 0; t < (SAME ? 1 : NT); ++t)
#pragma unroll
                for (int mb = 0; mb < B3; ++mb) best[t][mb] = f4{0.f, 0.f, 0.f, 0.f};   // outputs are >= 0 after the ReLU
        }
        f4 in[NT];
        bool live[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const Where wq = where(unit, sub, t);
            live[t] = wq.live;
            const int nb = nb_next[t];
            const int b = (unsigned)wq.ctr / (unsigned)a.m;
            const float *c3 = a.new_xyz + (size_t)wq.ctr * 3;
            const size_t src = (size_t)b * a.n + nb;
            const float *p3 = a.xyz + src * 3;
            const float rel[3] = {p3[0] - c3[0], p3[1] - c3[1], p3[2] - c3[2]};   // pointnet2_utils.py:252
            float v[4];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const int c = 4 * g + s_, e = c - a.cin;
                v[s_] = c < a.cin ? a.feat[src * a.cin + c] : e == 0 ? rel[0] : e == 1 ? rel[1] : e == 2 ? rel[2] : 0.0f;
            }
            in[t] = f4{v[0], v[1], v[2], v[3]};
        }
        {
            const bool wrap = sub + 1 == nsub;
            const int nunit = wrap ? unit + nwaves : unit, nsub_ = wrap ? 0 : sub + 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const Where wq = where(nunit, nsub_, t);
                nb_next[t] = a.idx[wq.ctr * a.ns + wq.tic * 16 + pos];
            }
        }
        {
            f4 h1[NT][B1], h2[NT][B2];
#pragma unroll
            for (int mb = 0; mb < B1; ++mb) {
                const f4 bi = *reinterpret_cast<const f4 *>(bias1 + 16 * mb);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f4 acc = bi;
                    PDM_CHAIN(acc, A1[mb], in[t])
                    h1[t][mb] = floor4(acc, 0.0f);
                }
            }
#pragma unroll
            for (int mb = 0; mb < B2; ++mb) {
                const f4 bi = *reinterpret_cast<const f4 *>(bias2 + 16 * mb);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f4 acc = bi;
#pragma unroll
                    for (int kb = 0; kb < B1; ++kb) { PDM_CHAIN(acc, A2[mb][kb], h1[t][kb]) }
                    h2[t][mb] = floor4(acc, 0.0f);
                }
            }
#pragma unroll
            for (int mb = 0; mb < B3; ++mb) {
                const f4 bi = *reinterpret_cast<const f4 *>(bias3 + 16 * mb);
                f4 v[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f4 acc = bi;
#pragma unroll
                    for (int kb = 0; kb < B2; ++kb) { PDM_CHAIN(acc, A3[mb][kb], h2[t][kb]) }
                    v[t] = floor4(acc, 0.0f);
                    if (!live[t]) v[t] = f4{0.f, 0.f, 0.f, 0.f};   // wave-uniform
                }
                if constexpr (SAME) {   // both tiles belong to one centre: pool them together
                    f4 u = v[0];
                    u.x = fmaxf(u.x, v[1].x); u.y = fmaxf(u.y, v[1].y); u.z = fmaxf(u.z, v[1].z); u.w = fmaxf(u.w, v[1].w);
                    u = row16_max_nonneg4(u);
                    best[0][mb].x = fmaxf(best[0][mb].x, u.x); best[0][mb].y = fmaxf(best[0][mb].y, u.y);
                    best[0][mb].z = fmaxf(best[0][mb].z, u.z); best[0][mb].w = fmaxf(best[0][mb].w, u.w);
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f4 u = row16_max_nonneg4(v[t]);
                        best[t][mb].x = fmaxf(best[t][mb].x, u.x); best[t][mb].y = fmaxf(best[t][mb].y, u.y);
                        best[t][mb].z = fmaxf(best[t][mb].z, u.z); best[t][mb].w = fmaxf(best[t][mb].w, u.w);
                    }
                }
            }
        }
        if (sub != nsub - 1) continue;
        // lane pos == 0 of every DPP row holds the row maxima of channels 16mb + 4g .. +3
#pragma unroll
        for (int t = 0; t < (SAME ? 1 : NT); ++t) {
            const int ctr = SAME ? unit : unit * cpu_ + t / tpc;
            if (ctr >= ncentres || pos != 0) continue;
            float *orow = a.out + (size_t)ctr * a.out_stride + a.out_coff;
#pragma unroll
            for (int mb = 0; mb < B3; ++mb) {
                const f4 v = best[t][mb];
                const int c0 = 16 * mb + 4 * g;
                if (c0 + 4 <= a.cout) {
                    *reinterpret_cast<f4 *>(orow + c0) = v;
                } else {
                    if (c0 < a.cout) orow[c0] = v.x;
                    if (c0 + 1 < a.cout) orow[c0 + 1] = v.y;
                    if (c0 + 2 < a.cout) orow[c0 + 2] = v.z;
                }
            }
        }
    }
#undef PDM_CHAIN
}

// Register-resident form over the compacted row list (sa_pack.hip): one wave = one aligned tile pair per step.
template <int B1, int B2, int B3>
__global__ __launch_bounds__(256) void sa_reg_packed_kernel(SaArgs a, const int2 *__restrict__ pack,
                                                            const int *__restrict__ meta,
                                                            const float *__restrict__ wpack,
                                                            const float *__restrict__ bias) {
    constexpr int NT = 2;
    const int lane = threadIdx.x & 63, pos = lane & 15, g = lane >> 4;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int m1 = meta[1], m2 = meta[2], m3 = meta[3], m4 = meta[4], m5 = meta[5];
    const int npairs = meta[6] >> 5;
    if (wave_id >= npairs) return;
    const f4 *__restrict__ w1 = reinterpret_cast<const f4 *>(wpack) + lane;
    const f4 *__restrict__ w2 = w1 + 64 * B1;
    const f4 *__restrict__ w3 = w2 + 64 * B2 * B1;
    f4 A1[B1], A2[B2][B1], A3[B3][B2];
#pragma unroll
    for (int mb = 0; mb < B1; ++mb) A1[mb] = w1[64 * mb];
#pragma unroll
    for (int mb = 0; mb < B2; ++mb)
#pragma unroll
        for (int kb = 0; kb < B1; ++kb) A2[mb][kb] = w2[64 * (mb * B1 + kb)];
#pragma unroll
    for (int mb = 0; mb < B3; ++mb)
#pragma unroll
        for (int kb = 0; kb < B2; ++kb) A3[mb][kb] = w3[64 * (mb * B2 + kb)];
    const float *__restrict__ bias1 = bias + 4 * g, *__restrict__ bias2 = bias1 + 16 * B1, *__restrict__ bias3 = bias2 + 16 * B2;
#define PDM_CHAIN(ACC, A, B)                                                          \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).x, (B).x, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).y, (B).y, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).z, (B).z, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A).w, (B).w, ACC, 0, 0, 0);
    int2 e_next[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) e_next[t] = pack[(wave_id * 2 + t) * 16 + pos];
    for (int unit = wave_id; unit < npairs; unit += nwaves) {
        f4 in[NT];
        int ctr[NT], L[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int2 e = e_next[t];
            ctr[t] = e.y;
            L[t] = pack_tile_L((unit * 2 + t) * 16, m1, m2, m3, m4, m5);
            const float *c3 = a.new_xyz + (size_t)(e.y >= 0 ? e.y : 0) * 3;
            const size_t src = (size_t)e.x;
            const float *p3 = a.xyz + src * 3;
            const float rel[3] = {p3[0] - c3[0], p3[1] - c3[1], p3[2] - c3[2]};   // pointnet2_utils.py:252
            float v[4];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const int c = 4 * g + s_, e_ = c - a.cin;
                v[s_] = c < a.cin ? a.feat[src * a.cin + c] : e_ == 0 ? rel[0] : e_ == 1 ? rel[1] : e_ == 2 ? rel[2] : 0.0f;
            }
            in[t] = f4{v[0], v[1], v[2], v[3]};
        }
        {
            const int nunit = unit + nwaves;
#pragma unroll
            for (int t = 0; t < NT; ++t) e_next[t] = pack[nunit < npairs ? (nunit * 2 + t) * 16 + pos : 0];
        }
        f4 h1[NT][B1], h2[NT][B2];
#pragma unroll
        for (int mb = 0; mb < B1; ++mb) {
            const f4 bi = *reinterpret_cast<const f4 *>(bias1 + 16 * mb);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f4 acc = bi;
                PDM_CHAIN(acc, A1[mb], in[t])
                h1[t][mb] = floor4(acc, 0.0f);
            }
        }
#pragma unroll
        for (int mb = 0; mb < B2; ++mb) {
            const f4 bi = *reinterpret_cast<const f4 *>(bias2 + 16 * mb);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f4 acc = bi;
#pragma unroll
                for (int kb = 0; kb < B1; ++kb) { PDM_CHAIN(acc, A2[mb][kb], h1[t][kb]) }
                h2[t][mb] = floor4(acc, 0.0f);
            }
        }
        const bool pair = L[0] >= 32;   // wave-uniform: both tiles belong to one centre
        float *orow[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int Lw = L[t] < 16 ? L[t] : 16;
            const bool writer = ctr[t] >= 0 && (pos & (Lw - 1)) == 0 && !(pair && t == 1);
            orow[t] = writer ? a.out + (size_t)ctr[t] * a.out_stride + a.out_coff : nullptr;
        }
#pragma unroll
        for (int mb = 0; mb < B3; ++mb) {
            const f4 bi = *reinterpret_cast<const f4 *>(bias3 + 16 * mb);
            f4 v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f4 acc = bi;
#pragma unroll
                for (int kb = 0; kb < B2; ++kb) { PDM_CHAIN(acc, A3[mb][kb], h2[t][kb]) }
                v[t] = floor4(acc, 0.0f);
            }
            if (pair) {
                v[0].x = fmaxf(v[0].x, v[1].x); v[0].y = fmaxf(v[0].y, v[1].y);
                v[0].z = fmaxf(v[0].z, v[1].z); v[0].w = fmaxf(v[0].w, v[1].w);
                v[0] = seg_max_nonneg4(v[0], 16);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] = seg_max_nonneg4(v[t], L[t]);
            }
            const int c0 = 16 * mb + 4 * g;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (!orow[t]) continue;
                if (c0 + 4 <= a.cout) {
                    *reinterpret_cast<f4 *>(orow[t] + c0) = v[t];
                } else {
                    if (c0 < a.cout) orow[t][c0] = v[t].x;
                    if (c0 + 1 < a.cout) orow[t][c0 + 1] = v[t].y;
                    if (c0 + 2 < a.cout) orow[t][c0 + 2] = v[t].z;
                }
            }
        }
    }
#undef PDM_CHAIN
}

int rows_gemm_launch(void *stream, int rows, int cin, const float *in_pm, int k0, int c1, const float *wpack,
                     const float *bias, int relu_last, float *out_pm, int out_stride, int cout);   // rows_gemm.hip
int rows_chain_launch(void *stream, int rows, int cin, const float *in_pm, int nlayers, const int *dims, const float *wpack,
                      const float *bias, int relu_last, float *out_pm, int out_stride, int cout, int *launched);   // rows_chain.hip
int rows_chain_pair_launch(void *stream, int rows, int cin, const float *in_pm, int nlayers, const int *dims, const float *wpack_a,
                           const float *bias_a, const float *wpack_b, const float *bias_b, int relu_last, float *out_a, int out_stride_a,
                           int cout_a, float *out_b, int out_stride_b, int cout_b, int *launched);   // rows_chain.hip
static int g_fused_chain = 1;       // 0: many-row MLPs go through the general chain kernel instead of rows_chain.hip
static int g_fused_pair = 1;        // 0: pdm_rows_mlp_fused_pair always runs its two chains as two launches
int fp_chain_launch(void *stream, int b, int n, int m, int c_skip, const float *z_pm, int z_stride, const float *skip_pm,
                    const int *idx, const float *weight, const int *dims, const float *wpack, const float *bias, float *out_pm,
                    int out_stride, int cout, int *launched);
int fp_pre_gemm_launch(void *stream, int b, int n, int m, int c_skip, const float *z_pm, int z_stride,
                       const float *skip_pm, const int *idx, const float *weight, int k1, int c2,
                       const float *wpack, const float *bias, float *out_pm, int out_stride, int cout);
static int g_fused_swz = 1;         // 0: round 1's LDS tile layout (rows padded by 4 floats, 2-way conflicts on b128 reads)
static int g_fused_gemm = 1;        // 0: single-layer rows go through the chain kernel instead of the LDS-tiled GEMM
static int g_fused_reg = 1;         // 0 switches the register-resident SA form off (A/B measurements)
static int g_fused_waves = 0;
static int g_fused_tiles = 0;
static int g_fused_groups = 0;      // 0 = auto
static int g_fused_lds_cap = 152 * 1024;  // two-tile form allowed up to this much LDS per workgroup (CU has 160 KB)
static int g_fused_wg_per_cu = 32;  // grid cap = 256 CUs x this many workgroups (grid-stride loop beyond)

static int fill_desc(const char *who, MlpDesc &d, int nlayers, const int *dims, int k0_real_max, int pool_floats,
                     long long ntiles, int *waves, int *tiles_per_wg, int *groups, bool two_tiles) {
    PDM_REQUIRE(nlayers >= 1 && nlayers <= FM_MAXL, PDM_E_BADARG, "%s: nlayers=%d not in [1,%d]", who, nlayers, FM_MAXL);
    PDM_REQUIRE(dims, PDM_E_BADARG, "%s: null dims", who);
    d.nlayers = nlayers;
    d.relu_last = 1;
    int wo = 0, bo = 0;
    for (int l = 0; l <= nlayers; ++l) {
        PDM_REQUIRE(dims[l] > 0 && dims[l] % 16 == 0, PDM_E_BADARG, "%s: padded width %d of level %d is not a positive multiple of 16", who, dims[l], l);
        d.K[l] = dims[l];
    }
    for (int l = 0; l < nlayers; ++l) {
        d.woff[l] = wo; d.boff[l] = bo;
        wo += d.K[l] * d.K[l + 1];
        bo += d.K[l + 1];
    }
    PDM_REQUIRE(d.K[0] >= k0_real_max, PDM_E_BADARG, "%s: padded input width %d < %d real channels", who, d.K[0], k0_real_max);
    // ping-pong buffers: Q holds outputs of layers 1,3; P holds the staged input and outputs of layer 2
    int q = 0, p = 0;
    for (int l = 1; l < nlayers; ++l) {
        if (l & 1) q = q > d.K[l] ? q : d.K[l]; else p = p > d.K[l] ? p : d.K[l];
    }
    // waves per workgroup: wide layers split their output blocks over more waves (more parallelism per
    // tile, LDS tile shared); narrow layers keep waves independent.
    int maxmb = 0;
    for (int l = 1; l <= nlayers; ++l) maxmb = maxmb > (d.K[l] >> 4) ? maxmb : (d.K[l] >> 4);
    const int W = g_fused_waves > 0 ? g_fused_waves : maxmb >= 24 ? 8 : maxmb >= 12 ? 4 : maxmb >= 6 ? 2 : 1;
    *waves = W;
    // stage the gathered input tile in LDS when the workgroup still fits >= 8 waves per CU (160 KB LDS)
    const int p_staged = p > d.K[0] ? p : d.K[0];
    // two 16-position tiles per workgroup share every weight fragment (half the A traffic, twice the MFMA
    // work per pass); keep one when the problem has few tiles or the LDS tiles would not fit 64 KB
    // two tiles per group share each weight fragment: measured -5 % on the wide SA layers, -7..-18 % on the
    // hoisted FP form, +8 % on the unhoisted FP form (heavier input provider) and up to 2x slower on plain rows
    // (few tiles); knob: 1 forces one tile, 2 forces two where instantiated
    int NT = (W >= 4 && (g_fused_tiles == 2 || (g_fused_tiles == 0 && two_tiles))) ? 2 : 1;
    (void)ntiles;
    // row width of an LDS tile: swizzled rows are whole 256-byte segments, the old layout pads by 4 floats
    auto roww = [](int k) { return g_fused_swz ? (k + 63) / 64 * 64 : k + 4; };
    if ((NT * 16 * (roww(p) + roww(q)) + pool_floats) * 4 > g_fused_lds_cap) NT = 1;
    *tiles_per_wg = NT;
    const int budget = (W == 1 ? 20 : W == 2 ? 40 : W == 4 ? 52 : 64) * 1024;
    d.stage_in = (NT * 16 * (roww(p_staged) + roww(q)) + pool_floats) * 4 <= budget ? 1 : 0;
    if (d.stage_in) p = p_staged;
    d.swz = g_fused_swz;
    d.lds_p = roww(p);
    d.lds_q = roww(q);
    d.pool_floats = pool_floats;
    // position-split groups per workgroup: as many as fit 8 waves and 64 KB of LDS
    const int region_bytes = (NT * 16 * (d.lds_p + d.lds_q) + pool_floats) * 4;
    // measured per launch (tools/diag/fused_calls.py): small workgroups win — W = 1: 1 group, W = 2: 2 groups
    // (FP1 -18 %, first rows call -15 % against 4), wide layers indifferent
    int G = NT == 2 ? 1 : (g_fused_groups > 0 ? g_fused_groups : W == 2 ? 2 : 1);
    while (G > 1 && (G * W > 8 || (long long)G * region_bytes > 64 * 1024 || (long long)G * NT * 4 > ntiles)) G >>= 1;
    *groups = G;
    return 0;
}

}  // namespace pdm

using namespace pdm;

// Tuning knob (not part of the reference-facing ABI): force the waves-per-workgroup choice (0 = auto).
extern "C" int pdm_tune_fused_waves(int w) { const int old = g_fused_waves; g_fused_waves = (w == 1 || w == 2 || w == 4 || w == 8) ? w : 0; return old; }

// more than 64 KB of dynamic LDS has to be granted per kernel function (once)
static void allow_lds(const void *fn, size_t bytes) {
    if (bytes > 64 * 1024) (void)grant_lds(fn, 160 * 1024);   // per function and per device (common.h)
}
extern "C" int pdm_tune_fused_lds_cap(int bytes) { const int old = g_fused_lds_cap; if (bytes >= 16 * 1024 && bytes <= 160 * 1024) g_fused_lds_cap = bytes; return old; }
extern "C" int pdm_tune_fused_swz(int on) { const int old = g_fused_swz; g_fused_swz = on != 0; return old; }
extern "C" int pdm_tune_fused_chain(int on) { const int old = g_fused_chain; g_fused_chain = on != 0; return old; }
extern "C" int pdm_tune_fused_pair(int on) { const int old = g_fused_pair; g_fused_pair = on != 0; return old; }
extern "C" int pdm_tune_fused_gemm(int on) { const int old = g_fused_gemm; g_fused_gemm = on != 0; return old; }
extern "C" int pdm_tune_fused_reg(int on) { const int old = g_fused_reg; g_fused_reg = on != 0; return old; }
extern "C" int pdm_tune_fused_wg_per_cu(int n) { const int old = g_fused_wg_per_cu; if (n > 0) g_fused_wg_per_cu = n; return old; }
extern "C" int pdm_tune_fused_groups(int n) { const int old = g_fused_groups; g_fused_groups = (n == 1 || n == 2 || n == 4 || n == 8) ? n : 0; return old; }
extern "C" int pdm_tune_fused_tiles(int t) { const int old = g_fused_tiles; g_fused_tiles = t; return old; }  // 0 = heuristic, 1 / 2 = force tiles per group

#define FUSED_LAUNCH2(KERNEL, W, G, PRE, blocks, lds_bytes, ...)                                              \
    do {                                                                                                       \
        constexpr int MAXNB = (W <= 2) ? 2 : 4; /* narrow layers: fewer registers, more waves in flight */     \
        if (NT == 2 && W >= 4 && G == 1) { /* two tiles share each weight fragment */                          \
            allow_lds(reinterpret_cast<const void *>(&KERNEL<(W >= 4 ? W : 4), 2, 2, 1, PRE>), lds_bytes);     \
            hipLaunchKernelGGL((KERNEL<(W >= 4 ? W : 4), 2, 2, 1, PRE>), dim3(blocks), dim3(64 * W * G),       \
                               lds_bytes, as_stream(stream), __VA_ARGS__);                                     \
        } else {                                                                                               \
            allow_lds(reinterpret_cast<const void *>(&KERNEL<W, 1, MAXNB, G, PRE>), lds_bytes);                \
            hipLaunchKernelGGL((KERNEL<W, 1, MAXNB, G, PRE>), dim3(blocks), dim3(64 * W * G), lds_bytes,       \
                               as_stream(stream), __VA_ARGS__);                                                \
        }                                                                                                      \
    } while (0)
#define FUSED_LAUNCH1(KERNEL, W, G, blocks, lds_bytes, ...)                                                   \
    do {                                                                                                       \
        if (pre_form) FUSED_LAUNCH2(KERNEL, W, G, true, blocks, lds_bytes, __VA_ARGS__);                       \
        else FUSED_LAUNCH2(KERNEL, W, G, false, blocks, lds_bytes, __VA_ARGS__);                               \
    } while (0)

// (W channel-split waves, G position groups), W * G <= 8; NT = 1 (NT = 2 measured no faster)
#define FUSED_DISPATCH(KERNEL, W, G, blocks, lds_bytes, ...)                                                   \
    do {                                                                                                       \
        const int key = W * 16 + G;                                                                            \
        if (key == 1 * 16 + 1) FUSED_LAUNCH1(KERNEL, 1, 1, blocks, lds_bytes, __VA_ARGS__);                    \
        else if (key == 1 * 16 + 2) FUSED_LAUNCH1(KERNEL, 1, 2, blocks, lds_bytes, __VA_ARGS__);               \
        else if (key == 1 * 16 + 4) FUSED_LAUNCH1(KERNEL, 1, 4, blocks, lds_bytes, __VA_ARGS__);               \
        else if (key == 1 * 16 + 8) FUSED_LAUNCH1(KERNEL, 1, 8, blocks, lds_bytes, __VA_ARGS__);               \
        else if (key == 2 * 16 + 1) FUSED_LAUNCH1(KERNEL, 2, 1, blocks, lds_bytes, __VA_ARGS__);               \
        else if (key == 2 * 16 + 2) FUSED_LAUNCH1(KERNEL, 2, 2, blocks, lds_bytes, __VA_ARGS__);               \
        else if (key == 2 * 16 + 4) FUSED_LAUNCH1(KERNEL, 2, 4, blocks, lds_bytes, __VA_ARGS__);               \
        else if (key == 4 * 16 + 1) FUSED_LAUNCH1(KERNEL, 4, 1, blocks, lds_bytes, __VA_ARGS__);               \
        else if (key == 4 * 16 + 2) FUSED_LAUNCH1(KERNEL, 4, 2, blocks, lds_bytes, __VA_ARGS__);               \
        else FUSED_LAUNCH1(KERNEL, 8, 1, blocks, lds_bytes, __VA_ARGS__);                                      \
    } while (0)

// Everything sa_fused_launch decides before launching: argument checks, layer descriptor, kernel arguments, workgroup shape.
struct SaPlan {
    MlpDesc d;
    SaArgs a;
    int W, NT, G, blocks;
    size_t lds_bytes;
    bool pre_form, packed, reg;     // reg: the register-resident kernels (small scales) apply
    int b1, b2, b3;
    long long cap_pairs;
    const int2 *pack2;
};
static int sa_plan(int b, int n, int m, int cin, int nsample, const float *xyz, const float *new_xyz, const float *feat_pm,
                   const float *z_pm, int z_stride, int z_coff, const int *idx, int nlayers, const int *dims, const float *wpack,
                   const float *bias, float *out_pm, int out_stride, int out_coff, int cout, const int *pack, const int *meta, SaPlan &pl) {
    PDM_REQUIRE(b >= 0 && n >= 1 && m >= 0 && cin >= 0 && nsample > 0, PDM_E_BADARG, "sa_mlp_fused: bad size");
    PDM_REQUIRE(nsample % 16 == 0, PDM_E_BADARG, "sa_mlp_fused: nsample=%d must be a multiple of 16", nsample);
    pl.blocks = 0;
    if (b == 0 || m == 0) return 0;
    PDM_REQUIRE(xyz && new_xyz && (idx || (pack && meta)) && wpack && bias && out_pm && (cin == 0 || feat_pm), PDM_E_BADARG,
                "sa_mlp_fused: null pointer");
    const bool packed = pack != nullptr;
    if (packed) {
        PDM_REQUIRE(nsample == 16 || nsample == 32, PDM_E_BADARG, "sa_mlp_packed: nsample=%d (16 or 32)", nsample);
        PDM_REQUIRE((long long)b * m * nsample + 256 < (1ll << 31) && (reinterpret_cast<uintptr_t>(pack) & 7) == 0,
                    PDM_E_BADARG, "sa_mlp_packed: row list too long or misaligned");
    }
    pl.packed = packed;
    pl.pack2 = reinterpret_cast<const int2 *>(pack);
    pl.cap_pairs = (long long)pdm_sa_pack_rows(b, m, nsample) / 32;
    MlpDesc &d = pl.d;
    int W = 1, NT = 1, G = 1;
    const long long sa_tiles = (long long)b * m * (nsample / 16);
    int rc = fill_desc("sa_mlp_fused", d, nlayers, dims, cin + 3,
                       dims ? dims[nlayers > 0 && nlayers <= FM_MAXL ? nlayers : 0] : 0, sa_tiles, &W, &NT, &G, true);
    if (rc) return rc;
    if (z_pm) {
        d.stage_in = 0;  // layer-1 input is the lane's own xyz offset: nothing to stage
        PDM_REQUIRE(z_stride % 4 == 0 && z_coff % 4 == 0 && z_coff >= 0 && z_coff + d.K[1] <= z_stride &&
                        (reinterpret_cast<uintptr_t>(z_pm) & 15) == 0,
                    PDM_E_BADARG, "sa_mlp_fused_pre: z rows need %d floats at offset %d of stride %d, 16-byte aligned", d.K[1], z_coff, z_stride);
    }
    PDM_REQUIRE(cout > 0 && cout <= d.K[nlayers] && out_coff >= 0 && out_coff + cout <= out_stride, PDM_E_BADARG,
                "sa_mlp_fused: cout=%d coff=%d stride=%d", cout, out_coff, out_stride);
    PDM_REQUIRE(out_stride % 4 == 0 && out_coff % 4 == 0 && (reinterpret_cast<uintptr_t>(out_pm) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(wpack) & 15) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0 &&
                    (cin % 4 != 0 || (reinterpret_cast<uintptr_t>(feat_pm) & 15) == 0),
                PDM_E_BADARG, "sa_mlp_fused: out/wpack/bias/feat must be 16-byte aligned, out_stride and out_coff multiples of 4");
    pl.a = SaArgs{b, n, m, cin, nsample, xyz, new_xyz, feat_pm, idx, out_pm, out_stride, out_coff, cout, z_pm, z_stride, z_coff};
    pl.pre_form = z_pm != nullptr;
    pl.lds_bytes = (size_t)G * (NT * 16 * (d.lds_p + d.lds_q) + d.K[nlayers]) * sizeof(float);
    PDM_REQUIRE(pl.lds_bytes <= 160 * 1024, PDM_E_TOOLARGE, "sa_mlp_fused: needs %zu bytes of LDS", pl.lds_bytes);
    pl.W = W; pl.NT = NT; pl.G = G;
    pl.b1 = d.K[1] >> 4; pl.b2 = d.K[2] >> 4; pl.b3 = d.K[3] >> 4;
    pl.reg = g_fused_reg && !pl.pre_form && nlayers == 3 && d.K[0] == 16 && (long long)b * m * nsample < (1ll << 31) &&
             (long long)b * n * (cin > 3 ? cin : 3) < (1ll << 31) &&
             ((pl.b1 == 1 && pl.b2 == 1 && pl.b3 == 2) || (pl.b1 == 2 && pl.b2 == 2 && pl.b3 == 4));
    const int tpc_ = nsample / 16;
    const long long units_ = packed ? pl.cap_pairs : tpc_ >= NT ? (long long)b * m : ((long long)b * m + NT / tpc_ - 1) / (NT / tpc_);
    const long long niter = (units_ + G - 1) / G;
    const long long cap = (long long)256 * g_fused_wg_per_cu;
    pl.blocks = (int)(niter < cap ? niter : cap);
    return 0;
}

static int sa_fused_launch(void *stream, int b, int n, int m, int cin, int nsample, const float *xyz,
                           const float *new_xyz, const float *feat_pm, const float *z_pm, int z_stride,
                           int z_coff, const int *idx, int nlayers, const int *dims, const float *wpack,
                           const float *bias, float *out_pm, int out_stride, int out_coff, int cout,
                           const int *pack = nullptr, const int *meta = nullptr) {
    SaPlan pl;
    if (int rc = sa_plan(b, n, m, cin, nsample, xyz, new_xyz, feat_pm, z_pm, z_stride, z_coff, idx, nlayers, dims, wpack, bias, out_pm,
                         out_stride, out_coff, cout, pack, meta, pl)) return rc;
    if (b == 0 || m == 0) return 0;
    const MlpDesc &d = pl.d;
    const SaArgs &a = pl.a;
    const int W = pl.W, NT = pl.NT, G = pl.G, blocks = pl.blocks;
    const bool pre_form = pl.pre_form, packed = pl.packed;
    const size_t lds_bytes = pl.lds_bytes;
    const int2 *pack2 = pl.pack2;
    const long long cap_pairs = pl.cap_pairs;
    if (pl.reg) {
        // small scales: the whole MLP stays in registers (weights, activations), one wave per tile pair
        const int b1 = pl.b1, b2 = pl.b2, b3 = pl.b3;
        const long long centres = (long long)b * m;
        const long long units = nsample >= 32 ? centres : (centres + 1) / 2;
        const long long want = (units + 3) / 4, capr = 256 * 8;
        const int blocks_r = (int)(want < capr ? want : capr);
#define PDM_REG_LAUNCH(X, Y, Z)                                                                                          \
    if (b1 == X && b2 == Y && b3 == Z) {                                                                                \
        if (packed) {                                                                                                   \
            const long long wantp = (cap_pairs + 3) / 4;                                                                \
            hipLaunchKernelGGL((sa_reg_packed_kernel<X, Y, Z>), dim3((int)(wantp < capr ? wantp : capr)), dim3(256), 0, \
                               as_stream(stream), a, pack2, meta, wpack, bias);                                         \
        } else if (nsample >= 32)                                                                                              \
            hipLaunchKernelGGL((sa_reg_mlp_kernel<X, Y, Z, true>), dim3(blocks_r), dim3(256), 0, as_stream(stream), a, wpack, bias);  \
        else                                                                                                            \
            hipLaunchKernelGGL((sa_reg_mlp_kernel<X, Y, Z, false>), dim3(blocks_r), dim3(256), 0, as_stream(stream), a, wpack, bias); \
        return check_launch("sa_mlp_fused(reg)");                                                                       \
    }
        PDM_REG_LAUNCH(1, 1, 2)
        PDM_REG_LAUNCH(2, 2, 4)
#undef PDM_REG_LAUNCH
    }
    if (packed) {
        FUSED_DISPATCH(sa_packed_fused_kernel, W, G, blocks, lds_bytes, d, a, pack2, meta, wpack, bias);
        return check_launch("sa_mlp_packed");
    }
    FUSED_DISPATCH(sa_mlp_fused_kernel, W, G, blocks, lds_bytes, d, a, wpack, bias);
    return check_launch("sa_mlp_fused");
}

extern "C" int pdm_sa_mlp_fused(void *stream, int b, int n, int m, int cin, int nsample,
                                const float *xyz, const float *new_xyz, const float *feat_pm,
                                const int *idx, int nlayers, const int *dims, const float *wpack,
                                const float *bias, float *out_pm, int out_stride, int out_coff,
                                int cout) {
    return sa_fused_launch(stream, b, n, m, cin, nsample, xyz, new_xyz, feat_pm, nullptr, 0, 0, idx, nlayers, dims,
                           wpack, bias, out_pm, out_stride, out_coff, cout);
}

// Same SA scale with the FEATURE part of layer 1 applied beforehand to the n source points
// (z = W1[:, features] f, pdm_rows_mlp_fused with relu_last = 0): layer 1 = relu(z[nb] + W1[:, xyz] (x_nb - c) + b).
// The contraction runs once per source point instead of once per (centre, neighbour) pair.
extern "C" int pdm_sa_mlp_fused_pre(void *stream, int b, int n, int m, int nsample, const float *xyz,
                                    const float *new_xyz, const float *z_pm, int z_stride, int z_coff,
                                    const int *idx, int nlayers, const int *dims, const float *wpack,
                                    const float *bias, float *out_pm, int out_stride, int out_coff, int cout) {
    PDM_REQUIRE(z_pm || b == 0 || m == 0, PDM_E_BADARG, "sa_mlp_fused_pre: null z");
    return sa_fused_launch(stream, b, n, m, 0, nsample, xyz, new_xyz, nullptr, z_pm, z_stride, z_coff, idx, nlayers,
                           dims, wpack, bias, out_pm, out_stride, out_coff, cout);
}

// The same SA scale over a compacted neighbour list (pdm_sa_pack): duplicate padding rows are not computed.
// z_pm null = unhoisted form (feat_pm rows, cin channels); non-null = hoisted form (feat_pm / cin ignored).
extern "C" int pdm_sa_mlp_packed(void *stream, int b, int n, int m, int cin, int nsample, const float *xyz,
                                 const float *new_xyz, const float *feat_pm, const float *z_pm, int z_stride,
                                 int z_coff, const int *pack, const int *meta, int nlayers, const int *dims,
                                 const float *wpack, const float *bias, float *out_pm, int out_stride, int out_coff,
                                 int cout) {
    PDM_REQUIRE((pack && meta) || b == 0 || m == 0, PDM_E_BADARG, "sa_mlp_packed: null row list");
    return sa_fused_launch(stream, b, n, m, z_pm ? 0 : cin, nsample, xyz, new_xyz, z_pm ? nullptr : feat_pm, z_pm,
                           z_stride, z_coff, nullptr, nlayers, dims, wpack, bias, out_pm, out_stride, out_coff, cout,
                           pack, meta);
}

// Both scales of an SA level over their compacted lists in ONE launch where they map to the same kernel instantiation (the deep
// levels of PointNet2MSG do; see sa_packed_pair_kernel), otherwise two pdm_sa_mlp_packed launches.  Per-scale arguments come as
// arrays of two; xyz / new_xyz / feat_pm / z_pm / out_pm are the level's.  Bit-identical to the two calls.
static int g_sa_pair = 1;
extern "C" int pdm_tune_sa_pair(int on) { const int old = g_sa_pair; g_sa_pair = on != 0; return old; }
extern "C" int pdm_sa_mlp_packed_pair(void *stream, int b, int n, int m, int cin, const int *nsample, const float *xyz,
                                      const float *new_xyz, const float *feat_pm, const float *z_pm, int z_stride, const int *z_coff,
                                      const int *const *pack, const int *const *meta, const int *nlayers, const int *const *dims,
                                      const float *const *wpack, const float *const *bias, float *out_pm, int out_stride,
                                      const int *out_coff, const int *cout) {
    PDM_REQUIRE(nsample && z_coff && pack && meta && nlayers && dims && wpack && bias && out_coff && cout, PDM_E_BADARG,
                "sa_mlp_packed_pair: null table");
    PDM_REQUIRE((pack[0] && meta[0] && pack[1] && meta[1]) || b == 0 || m == 0, PDM_E_BADARG, "sa_mlp_packed_pair: null row list");
    SaPlan pl[2];
    for (int k = 0; k < 2; ++k)
        if (int rc = sa_plan(b, n, m, z_pm ? 0 : cin, nsample[k], xyz, new_xyz, z_pm ? nullptr : feat_pm, z_pm, z_stride, z_coff[k], nullptr,
                             nlayers[k], dims[k], wpack[k], bias[k], out_pm, out_stride, out_coff[k], cout[k], pack[k], meta[k], pl[k])) return rc;
    if (b == 0 || m == 0) return 0;
    const bool same = g_sa_pair && !pl[0].reg && !pl[1].reg && pl[0].W == pl[1].W && pl[0].NT == pl[1].NT && pl[0].G == pl[1].G &&
                      pl[0].pre_form == pl[1].pre_form && pl[0].packed && pl[1].packed;
    if (!same) {
        for (int k = 0; k < 2; ++k)
            if (int rc = pdm_sa_mlp_packed(stream, b, n, m, cin, nsample[k], xyz, new_xyz, feat_pm, z_pm, z_stride, z_coff[k], pack[k], meta[k],
                                           nlayers[k], dims[k], wpack[k], bias[k], out_pm, out_stride, out_coff[k], cout[k])) return rc;
        return 0;
    }
    const int W = pl[0].W, NT = pl[0].NT, G = pl[0].G;
    const bool pre_form = pl[0].pre_form;
    const size_t lds_bytes = pl[0].lds_bytes > pl[1].lds_bytes ? pl[0].lds_bytes : pl[1].lds_bytes;
    const dim3 blocks((unsigned)(pl[0].blocks > pl[1].blocks ? pl[0].blocks : pl[1].blocks), 2);
    const SaScale s0{pl[0].d, pl[0].a, pl[0].pack2, meta[0], wpack[0], bias[0]}, s1{pl[1].d, pl[1].a, pl[1].pack2, meta[1], wpack[1], bias[1]};
    FUSED_DISPATCH(sa_packed_pair_kernel, W, G, blocks, lds_bytes, s0, s1);
    return check_launch("sa_mlp_packed_pair");
}

// mode 0: FP module (known rows interpolated in the kernel); 1: FP module with pre-projected known rows z;
// 2: plain rows (no interpolation: c_known = 0, the rows are `skip`)
static int fp_fused_launch(void *stream, int mode, int relu_last, int b, int n, int m, int c_known, int c_skip,
                           const float *known_pm, const float *z_pm, int z_stride, const float *skip_pm,
                           const int *idx, const float *weight, int nlayers, const int *dims,
                           const float *wpack, const float *bias, float *out_pm, int out_stride, int cout) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 1 && c_known >= 0 && c_skip >= 0 && c_known + c_skip >= (mode == 1 ? 0 : 1),
                PDM_E_BADARG, "fp_mlp_fused: bad size");
    if (b == 0 || n == 0) return 0;
    PDM_REQUIRE(wpack && bias && out_pm && (c_skip == 0 || skip_pm) && (mode == 2 || (idx && weight)) &&
                    (mode != 0 || known_pm) && (mode != 1 || z_pm),
                PDM_E_BADARG, "fp_mlp_fused: null pointer");
    MlpDesc d;
    int W = 1, NT = 1, G = 1;
    const long long ntiles = (long long)b * ((n + 15) / 16);
    int rc = fill_desc("fp_mlp_fused", d, nlayers, dims, c_known + c_skip, 0, ntiles, &W, &NT, &G, mode == 1);
    if (rc) return rc;
    d.relu_last = relu_last ? 1 : 0;
    if (mode == 1)
        PDM_REQUIRE(z_stride % 4 == 0 && z_stride >= d.K[1] && (reinterpret_cast<uintptr_t>(z_pm) & 15) == 0, PDM_E_BADARG,
                    "fp_mlp_fused_pre: z rows need %d floats, stride %d, 16-byte aligned", d.K[1], z_stride);
    PDM_REQUIRE(cout > 0 && cout <= d.K[nlayers] && cout <= out_stride, PDM_E_BADARG, "fp_mlp_fused: cout=%d stride=%d", cout, out_stride);
    PDM_REQUIRE(out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out_pm) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(wpack) & 15) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0 &&
                    (c_known % 4 != 0 || (reinterpret_cast<uintptr_t>(known_pm) & 15) == 0) &&
                    (c_skip % 4 != 0 || c_skip == 0 || (reinterpret_cast<uintptr_t>(skip_pm) & 15) == 0),
                PDM_E_BADARG, "fp_mlp_fused: buffers must be 16-byte aligned and out_stride a multiple of 4");
    FpArgs a{b, n, m, c_known, c_skip, known_pm, skip_pm, idx, weight, out_pm, out_stride, cout, mode == 1 ? z_pm : nullptr, z_stride};
    const bool pre_form = mode == 1;
    PDM_REQUIRE(!pre_form || (long long)m * z_stride < (1ll << 31), PDM_E_TOOLARGE, "fp_mlp_fused_pre: m * z_stride overflows 32-bit row offsets");
    if (g_fused_chain && pre_form && nlayers == 2 && z_stride >= d.K[1] &&
        (c_skip <= 4 || (c_skip % 4 == 0 && (reinterpret_cast<uintptr_t>(skip_pm) & 15) == 0))) {
        // many rows, widths that fit the register-resident chain (rows_chain.hip): FP1 and FP2 of the bench model
        int launched = 0;
        const int crc = fp_chain_launch(stream, b, n, m, c_skip, z_pm, z_stride, skip_pm, idx, weight, dims, wpack, bias, out_pm,
                                        out_stride, cout, &launched);
        if (crc || launched) return crc;
    }
    if (g_fused_gemm && pre_form && nlayers == 2 && c_skip <= 4 && d.K[0] == 16 && d.K[1] >= 64 && d.K[2] >= 64 &&
        (long long)b * n >= 32768 && (long long)b * n < (1ll << 31) && z_stride >= d.K[1])
        // wide second layer over many rows: LDS-tiled GEMM, layer 1 made on the fly while staging (rows_gemm.hip)
        return fp_pre_gemm_launch(stream, b, n, m, c_skip, z_pm, z_stride, skip_pm, idx, weight, d.K[1], d.K[2], wpack, bias,
                                  out_pm, out_stride, cout);
    const size_t lds_bytes = (size_t)G * (NT * 16 * (d.lds_p + d.lds_q)) * sizeof(float);
    PDM_REQUIRE(lds_bytes <= 160 * 1024, PDM_E_TOOLARGE, "fp_mlp_fused: needs %zu bytes of LDS", lds_bytes);
    const long long niter = ((ntiles + NT - 1) / NT + G - 1) / G;
    const long long cap = (long long)256 * g_fused_wg_per_cu;
    const int blocks = (int)(niter < cap ? niter : cap);
    FUSED_DISPATCH(fp_mlp_fused_kernel, W, G, blocks, lds_bytes, d, a, wpack, bias);
    return check_launch("fp_mlp_fused");
}

extern "C" int pdm_fp_mlp_fused(void *stream, int b, int n, int m, int c_known, int c_skip,
                                const float *known_pm, const float *skip_pm, const int *idx,
                                const float *weight, int nlayers, const int *dims, const float *wpack,
                                const float *bias, float *out_pm, int out_stride, int cout) {
    PDM_REQUIRE(c_known >= 1, PDM_E_BADARG, "fp_mlp_fused: c_known=%d", c_known);
    return fp_fused_launch(stream, 0, 1, b, n, m, c_known, c_skip, known_pm, nullptr, 0, skip_pm, idx, weight, nlayers,
                           dims, wpack, bias, out_pm, out_stride, cout);
}

// Same FP module with the KNOWN-feature part of layer 1 applied beforehand to the m known points
// (z = W1[:, known] f, pdm_rows_mlp_fused with relu_last = 0).  Interpolation is linear, so
// layer 1 = relu(sum_k w_k z[idx_k] + W1[:, skip] s + b): the wide contraction runs over m rows instead of n.
// dims[0] = padded skip width (16 with all-zero weights when c_skip == 0).
extern "C" int pdm_fp_mlp_fused_pre(void *stream, int b, int n, int m, int c_skip, const float *z_pm,
                                    int z_stride, const float *skip_pm, const int *idx, const float *weight,
                                    int nlayers, const int *dims, const float *wpack, const float *bias,
                                    float *out_pm, int out_stride, int cout) {
    return fp_fused_launch(stream, 1, 1, b, n, m, 0, c_skip, nullptr, z_pm, z_stride, skip_pm, idx, weight, nlayers, dims,
                           wpack, bias, out_pm, out_stride, cout);
}

// Per-row MLP on fp32 MFMA: out[r] = act(W_L ... relu(W_1 in[r] + b_1) ... + b_L), rows (R, cin) contiguous.
// relu_last = 0 leaves the last layer linear (the pre-projections above, 1x1 convolutions without activation).
extern "C" int pdm_rows_mlp_fused(void *stream, int rows, int cin, const float *in_pm, int nlayers,
                                  const int *dims, const float *wpack, const float *bias, int relu_last,
                                  float *out_pm, int out_stride, int cout) {
    PDM_REQUIRE(rows >= 0 && cin >= 1, PDM_E_BADARG, "rows_mlp_fused: rows=%d cin=%d", rows, cin);
    if (g_fused_chain && dims && in_pm && wpack && bias && out_pm && cout > 0 && nlayers >= 1 && nlayers <= 3 && cout <= dims[nlayers] &&
        cout <= out_stride && out_stride % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(out_pm) | reinterpret_cast<uintptr_t>(wpack) | reinterpret_cast<uintptr_t>(bias) |
          reinterpret_cast<uintptr_t>(in_pm)) & 15) == 0) {
        // many rows, widths that fit the register-resident chain (rows_chain.hip): the hybrid head's MLPs
        int launched = 0;
        const int rc = rows_chain_launch(stream, rows, cin, in_pm, nlayers, dims, wpack, bias, relu_last, out_pm, out_stride, cout,
                                         &launched);
        if (rc || launched) return rc;
    }
    if (g_fused_gemm && nlayers == 1 && dims && dims[0] % 16 == 0 && dims[1] % 16 == 0 && dims[0] >= cin && dims[0] >= 32 &&
        dims[1] >= 64 && (long long)((rows + 127) / 128) * ((dims[1] + 127) / 128) >= 256 && in_pm && wpack && bias && out_pm && cout > 0 && cout <= dims[1] && cout <= out_stride &&
        out_stride % 4 == 0 && ((reinterpret_cast<uintptr_t>(out_pm) | reinterpret_cast<uintptr_t>(wpack) |
                                 reinterpret_cast<uintptr_t>(bias)) & 15) == 0)
        return rows_gemm_launch(stream, rows, cin, in_pm, dims[0], dims[1], wpack, bias, relu_last, out_pm, out_stride, cout);
    return fp_fused_launch(stream, 2, relu_last, 1, rows, 1, 0, cin, nullptr, nullptr, 0, in_pm, nullptr, nullptr, nlayers,
                           dims, wpack, bias, out_pm, out_stride, cout);
}

// Two per-row MLPs of EQUAL widths over the SAME rows (the point head's class and box stacks: one module with two
// make_fc_layers chains on one input, /root/reference/pcdet/models/dense_heads/point_head_box.py:7-60, :85-86):
//   out_a[r] = MLP_a(in[r]),  out_b[r] = MLP_b(in[r]);  dims / packing as pdm_rows_mlp_fused, cout_* <= dims[nlayers].
// One launch where rows_chain.hip has an instantiation (the rows are read once), otherwise two pdm_rows_mlp_fused calls;
// the results are bit-identical either way.
extern "C" int pdm_rows_mlp_fused_pair(void *stream, int rows, int cin, const float *in_pm, int nlayers, const int *dims,
                                       const float *wpack_a, const float *bias_a, const float *wpack_b, const float *bias_b,
                                       int relu_last, float *out_a, int out_stride_a, int cout_a, float *out_b, int out_stride_b,
                                       int cout_b) {
    PDM_REQUIRE(rows >= 0 && cin >= 1, PDM_E_BADARG, "rows_mlp_fused_pair: rows=%d cin=%d", rows, cin);
    if (g_fused_chain && g_fused_pair && dims && in_pm && wpack_a && bias_a && wpack_b && bias_b && out_a && out_b && out_a != out_b &&
        nlayers == 3 && cout_a > 0 && cout_b > 0 && cout_a <= dims[nlayers] && cout_b <= dims[nlayers] && cout_a <= out_stride_a &&
        cout_b <= out_stride_b && out_stride_a % 4 == 0 && out_stride_b % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(out_a) | reinterpret_cast<uintptr_t>(out_b) | reinterpret_cast<uintptr_t>(wpack_a) |
          reinterpret_cast<uintptr_t>(wpack_b) | reinterpret_cast<uintptr_t>(bias_a) | reinterpret_cast<uintptr_t>(bias_b) |
          reinterpret_cast<uintptr_t>(in_pm)) & 15) == 0) {
        int launched = 0;
        const int rc = rows_chain_pair_launch(stream, rows, cin, in_pm, nlayers, dims, wpack_a, bias_a, wpack_b, bias_b, relu_last, out_a,
                                              out_stride_a, cout_a, out_b, out_stride_b, cout_b, &launched);
        if (rc || launched) return rc;
    }
    if (int rc = pdm_rows_mlp_fused(stream, rows, cin, in_pm, nlayers, dims, wpack_a, bias_a, relu_last, out_a, out_stride_a, cout_a)) return rc;
    return pdm_rows_mlp_fused(stream, rows, cin, in_pm, nlayers, dims, wpack_b, bias_b, relu_last, out_b, out_stride_b, cout_b);
}
