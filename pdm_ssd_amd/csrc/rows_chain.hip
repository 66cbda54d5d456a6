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
#ifndef DW_PIN
#define DW_PIN 1    // depthwise prologue: slices whose tap reads may be in flight together
#endif
#ifndef DW_SL
#define DW_SL 4     // depthwise prologue: 16-channel slices per LDS stage (1, 2, 4 or 8 of the 8)
#endif
typedef const __attribute__((address_space(4))) f4 *cf4c;   // constant address space: uniform addresses load through the scalar cache
typedef const __attribute__((address_space(1))) f4 *gf4c;   // a pointer the compiler must treat as global (no flat loads)

struct RowsChainArgs {
    int rows, in_stride;        // floats between input rows (= cin, a multiple of 16)
    const float *in;
    const float *wpack, *bias;  // packed as fused.py::pack_layer, layers back to back
    int woff[3], boff[3];       // float offsets of each layer
    float *out;
    int out_stride, cout, relu_last;
    // depthwise prologue (DW instantiations): a row is a cell of a channels-last (B, H, W, C) map and the chain's input
    // is relu(depthwise3x3(map)[cell] + shift) formed on the fly (bev_head.hip's kernel, same fma order)
    int dw_H, dw_W, dw_by_xcd;
    const float *dw_w, *dw_shift;   // (9, C) tap-major with the BatchNorm scale folded in, (C)
};

constexpr int RC_THREADS = 256;
#ifndef RC_AHEAD
#define RC_AHEAD 1   // rows_chain_kernel: request the next tile's input rows under the current tile's second layer
#endif
constexpr int RC_CHUNK_F4 = 16 * 64;   // 16 fragments x 64 lanes
#ifndef RC_DIAG
#define RC_DIAG 0   // timing builds (make diag-rc, results wrong): 1 = no barriers in the chunk loop, 2 = no weight stream
#endif

// chunk (mb0 .. mb0 + nmb - 1) x (kb0 .. kb0 + 3) of a layer with NKB k-blocks: thread t fetches lane t % 64 of
// fragments (mb0 + i, kb0 + t / 64)
// (nkg < 4: the layer's last, partial group of k-blocks — the waves past it fetch nothing)
__device__ __forceinline__ void rc_fetch(f4 (&r)[4], const f4 *__restrict__ w, int nkb, int mb0, int kb0, int nmb, int t, int nkg = 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // (the weight pointers pass through an empty asm in the tile loop, after which the compiler no longer knows they
        //  are global and emits flat loads; those count on lgkmcnt too, so every wait for an LDS fragment also waited
        //  for the L2 round trip of the chunk just requested: the cast makes them global loads again, vmcnt only)
#if RC_DIAG & 16   // timing build: every fetch reads the same L1-resident 4 KB
        if (i < nmb && (nkg == 4 || (t >> 6) < nkg)) r[i] = ((gf4c)(w + i * 64))[t & 63];
        if (true) continue;
#endif
        if (i < nmb && (nkg == 4 || (t >> 6) < nkg))
            r[i] = ((gf4c)(w + ((size_t)(mb0 + i) * nkb + kb0) * 64))[t];   // uniform base (SGPRs) + one lane offset
    }
}
// A layer with ONE output block over 16 k-blocks (the heads' last layer, 256 -> <= 16): its 16 fragments travel as one chunk,
// slot (i, kbi) = k-block 4 kbi + i, so the slots of one kbi are four consecutive k-blocks (one barrier for the layer
// instead of four chunks of 16 MFMAs with a barrier each: the layer took ~9k cycles of a tile's ~110k for 4 % of its MFMAs)
__device__ __forceinline__ void rc_fetch_k16(f4 (&r)[4], const f4 *__restrict__ w, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = ((gf4c)(w + (size_t)((t >> 6) * 4 + i) * 64))[t & 63];
}
constexpr int rc_next_nkb(int nkb, int nmb) { return (nkb == 16 && nmb == 1) ? -16 : nkb; }   // rc_layer's next_nkb: < 0 = that form
__device__ __forceinline__ void rc_stash(const f4 (&r)[4], f4 *buf, int nmb, int t) {
#if RC_DIAG & 8    // timing build: the chunk is fetched but not written to LDS
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nmb) asm volatile("" :: "v"(r[i]));
    return;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nmb) buf[i * 256 + t] = r[i];
}

#ifndef RC_GLDS
#define RC_GLDS 0   // 1: the weight chunks go global -> LDS directly (global_load_lds_dwordx4: no VGPR staging, no ds_write, 20 VGPRs
                    // fewer; hipcc places the four pieces at the chunk's start and one vmcnt(0) in front of its barrier, no wait in
                    // front of the fragment reads).  Same results, measured SLOWER at a settled clock: point head 0.813 / 0.816 ->
                    // 0.828 / 0.826 ms, heat-map cells chain 0.272 -> 0.287 ms (tools/diag/rows_chain_rate.py): the vmcnt(0) at every
                    // chunk barrier also drains the next tile's row prefetch, and a piece costs more issue time beside 16 ds_reads
                    // and 64 MFMAs than a global_load + ds_write pair.  Kept as a switch; the register-staged stream is the product.
#endif
#if RC_GLDS
typedef __attribute__((address_space(3))) void *rc_lds_vp;
// one 1 KB piece: lane l's 16 bytes from src_lane land at dst_wave + 16 l (dst_wave is wave-uniform)
__device__ __forceinline__ void rc_glds(const f4 *src_lane, f4 *dst_wave) {
    __builtin_amdgcn_global_load_lds((gf4c)src_lane, (rc_lds_vp)(uintptr_t)(unsigned)(uintptr_t)dst_wave, 16, 0, 0);
}
__device__ __forceinline__ void rc_fetch_lds(f4 *dst, const f4 *__restrict__ w, int nkb, int mb0, int kb0, int nmb, int t, int nkg = 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nmb && (nkg == 4 || (t >> 6) < nkg)) rc_glds(w + ((size_t)(mb0 + i) * nkb + kb0) * 64 + t, dst + i * 256 + (t & ~63));
}
__device__ __forceinline__ void rc_fetch_k16_lds(f4 *dst, const f4 *__restrict__ w, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rc_glds(w + (size_t)((t >> 6) * 4 + i) * 64 + (t & 63), dst + i * 256 + (t & ~63));
}
#define RC_FETCH(r, dst, ...) rc_fetch_lds(dst, __VA_ARGS__)
#define RC_FETCH_K16(r, dst, w, t) rc_fetch_k16_lds(dst, w, t)
#define RC_STASH(r, dst, n, t) ((void)0)      /* the pieces are on their way; the barrier behind this point waits for them */
#else
#define RC_FETCH(r, dst, ...) rc_fetch(r, __VA_ARGS__)
#define RC_FETCH_K16(r, dst, w, t) rc_fetch_k16(r, w, t)
#define RC_STASH(r, dst, n, t) rc_stash(r, dst, n, t)
#endif

// One layer: in[NKB] -> acc[NMB] (bias added, floored).  `p` = which LDS buffer holds this layer's first chunk.
// next_*: the first chunk of the following layer (prefetched behind this layer's last chunk), next_w == nullptr: none.
// an[]: the A fragments of the NEXT k-block, read from LDS one k-block ahead of the MFMAs that use them (on entry: k-block
// 0 of this layer's first chunk; on exit: k-block 0 of the chunk prefetched last).  The chunk's barrier sits in front of
// its LAST k-block: the stash of chunk c + 1 and the barrier come there, then k-block 0 of chunk c + 1 is read from the
// other buffer under the last 16 MFMAs of chunk c, so no LDS read latency is exposed anywhere in the stream (with the
// barrier behind the chunk every k-block began with four ds_reads and a wait: the pipe sat idle ~1/4 of a wave's time).
// WAR on the buffers: all reads of chunk c (k-blocks 1.. during its k-blocks 0..) complete before the barrier of chunk c,
// and chunk c + 2 is stashed into that buffer only after it.
template <int NKB, int NMB>
__device__ __forceinline__ void rc_layer(const f4 (&in)[NKB], f4 (&acc)[NMB], const f4 *__restrict__ w, const float *__restrict__ bias,
                                         f4 *lds, int &p, float floor, const f4 *__restrict__ next_w, int next_nkb, int next_nmb,
                                         int t, int lane, f4 (&r)[4], f4 (&an)[4]) {
    constexpr int KG = (NKB + 3) / 4, MG = (NMB + 3) / 4, NCH = KG * MG;   // the last k-group of a layer may hold < 4 k-blocks
    const int g = lane >> 4;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) acc[mb] = *reinterpret_cast<const f4 *>(bias + 16 * mb + 4 * g);
    if constexpr (rc_next_nkb(NKB, NMB) < 0) {   // one chunk: slot (i, kbi) = k-block 4 kbi + i (rc_fetch_k16); k ascending as in the general form
        int fn = 0;
        if (!(RC_DIAG & 2) && next_w) {
            fn = next_nmb < 4 ? next_nmb : 4;
            RC_FETCH(r, lds + (p ^ 1) * RC_CHUNK_F4, next_w, next_nkb, 0, 0, fn, t, next_nkb < 4 ? next_nkb : 4);
        }
        const f4 *buf = lds + p * RC_CHUNK_F4 + lane, *nbuf = lds + (p ^ 1) * RC_CHUNK_F4 + lane;
#pragma unroll
        for (int kbi = 0; kbi < 4; ++kbi) {
            f4 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = an[i];
            if (kbi + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = buf[(i * 4 + kbi + 1) * 64];
            } else {
                if (fn) RC_STASH(r, lds + (p ^ 1) * RC_CHUNK_F4, fn, t);
#if !(RC_DIAG & 1)
                __syncthreads();
#endif
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = nbuf[(i * 4) * 64];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f4 b = in[4 * kbi + i];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b.x, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b.y, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b.z, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b.w, acc[0], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        p ^= 1;
    } else {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int mb0 = (c / KG) * 4, kb0 = (c % KG) * 4;
        const int nmb = NMB - mb0 < 4 ? NMB - mb0 : 4;
        const int nkg = NKB - kb0 < 4 ? NKB - kb0 : 4;
        // next chunk -> registers
        int fn = 0;
        if (RC_DIAG & 2) {
        } else if (c + 1 < NCH) {
            const int nm0 = ((c + 1) / KG) * 4, nk0 = ((c + 1) % KG) * 4;
            fn = NMB - nm0 < 4 ? NMB - nm0 : 4;
            RC_FETCH(r, lds + (p ^ 1) * RC_CHUNK_F4, w, NKB, nm0, nk0, fn, t, NKB - nk0 < 4 ? NKB - nk0 : 4);
        } else if (next_w && next_nkb < 0) {
            fn = 4;
            RC_FETCH_K16(r, lds + (p ^ 1) * RC_CHUNK_F4, next_w, t);
        } else if (next_w) {
            fn = next_nmb < 4 ? next_nmb : 4;
            RC_FETCH(r, lds + (p ^ 1) * RC_CHUNK_F4, next_w, next_nkb, 0, 0, fn, t, next_nkb < 4 ? next_nkb : 4);
        }
        // this chunk: nkg k-blocks x nmb output blocks
        const f4 *buf = lds + p * RC_CHUNK_F4 + lane, *nbuf = lds + (p ^ 1) * RC_CHUNK_F4 + lane;
#pragma unroll
        for (int kbi = 0; kbi < 4; ++kbi) {
            if (kbi >= nkg) break;
            f4 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = an[i];
            if (kbi + 1 < nkg) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nmb) an[i] = buf[(i * 4 + kbi + 1) * 64];
            } else {
                if (fn) RC_STASH(r, lds + (p ^ 1) * RC_CHUNK_F4, fn, t);
#if !(RC_DIAG & 1)
                __syncthreads();
#endif
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = nbuf[(i * 4) * 64];   // (fragments past the next chunk's nmb: unused values)
            }
            // (pinning these reads in front of the k-block's MFMAs with a sched_barrier measured 1 % slower than the scheduler's own
            //  placement behind the 14th MFMA)
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
            // keeps the fragment loads of later k-blocks from piling up in registers
            __builtin_amdgcn_sched_barrier(0);
        }
        p ^= 1;
    }
    }
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        acc[mb].x = __builtin_amdgcn_fmed3f(acc[mb].x, floor, __builtin_inff());
        acc[mb].y = __builtin_amdgcn_fmed3f(acc[mb].y, floor, __builtin_inff());
        acc[mb].z = __builtin_amdgcn_fmed3f(acc[mb].z, floor, __builtin_inff());
        acc[mb].w = __builtin_amdgcn_fmed3f(acc[mb].w, floor, __builtin_inff());
    }
}
// k-block 0 of the chunk in buffer p (the kernel prologue's first chunk)
__device__ __forceinline__ void rc_first_fragments(f4 (&an)[4], const f4 *lds, int p, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) an[i] = lds[p * RC_CHUNK_F4 + lane + (i * 4) * 64];
}

// NK0 = k-blocks (16 channels) of the input, NK1 .. NK3 = output blocks of layers 1 .. 3 (0 = layer absent)
#ifndef DW_WGS
#define DW_WGS 2    // depthwise form: workgroups per CU the register budget is set for
#endif
template <int NK0, int NK1, int NK2, int NK3, bool DW = false>
__global__ __launch_bounds__(RC_THREADS, DW ? DW_WGS : 2) void rows_chain_kernel(RowsChainArgs a) {
    __shared__ __attribute__((aligned(16))) f4 lds[2 * RC_CHUNK_F4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int pos = lane & 15, g = lane >> 4;
    const f4 *w1 = reinterpret_cast<const f4 *>(a.wpack + a.woff[0]);
    const f4 *w2 = NK2 ? reinterpret_cast<const f4 *>(a.wpack + a.woff[1]) : nullptr;
    const f4 *w3 = NK3 ? reinterpret_cast<const f4 *>(a.wpack + a.woff[2]) : nullptr;
    constexpr int NL = NK3 ? 3 : NK2 ? 2 : 1;
    const float neg_inf = -__builtin_inff();
    f4 r[4];
    __shared__ __attribute__((aligned(16))) f4 halo[DW ? DW_SL * 6 * 18 * 4 : 1];   // DW: DW_SL slices of (4 + 2) x (16 + 2) cells x 16 channels
    __shared__ __attribute__((aligned(16))) f4 xs[DW ? DW_SL * 64 * 5 : 1];         // DW: a stage's convolved quads, (slice, cell) rows of 4 quads + 1 pad
    const int ntx = DW ? (a.dw_W + 15) / 16 : 1, nty = DW ? (a.dw_H + 3) / 4 : 1;
    const long long ntiles = DW ? (long long)(a.rows / (a.dw_H * a.dw_W)) * nty * ntx : ((long long)a.rows + 63) / 64;
    // first chunk of layer 1 for the first tile; every later tile finds it prefetched behind the last chunk of the
    // tile before (the weights are the same for every tile), so the stream of chunks never stops at a tile boundary
    int p = 0;
    RC_FETCH(r, lds, w1, NK0, 0, 0, NK1 < 4 ? NK1 : 4, t);
    RC_STASH(r, lds, NK1 < 4 ? NK1 : 4, t);
    __syncthreads();
    f4 an[4];
    rc_first_fragments(an, lds, 0, lane);
    constexpr bool AHEAD = !DW && NL >= 2 && RC_AHEAD;
    f4 xn[AHEAD ? NK0 : 1];
    // DW: workgroups are dealt to the 8 XCDs round-robin, so with tiles walked in launch order the patches either side of a
    // patch (which share its halo) ran on other XCDs and every halo cell came from HBM again (976 MB of traffic for a 577 MB
    // map, exactly the 108 / 64 halo ratio).  XCD x walks the contiguous range [x per_xcd, (x + 1) per_xcd) of patches instead:
    // neighbours meet in the same L2.
    const bool by_xcd = DW && a.dw_by_xcd && (gridDim.x & 7) == 0;
    const long long per_xcd = (ntiles + 7) >> 3, nsteps = by_xcd ? per_xcd * 8 : ntiles;
    // DW: the halo of ALL slices of a patch sits in registers (hva, 64 VGPRs), requested one TILE ahead, under the
    // previous tile's chain (with the next slice requested under the current slice's taps, ~500 cycles of work against an
    // HBM round trip, every slice waited for memory)
    f4 hva[DW ? NK0 : 1][2];
    // (patch coordinates: two 32-bit divisions per tile, computed once — when the patch's halo is requested — and carried to
    //  the iteration that consumes it; the first form redid four 64-bit divisions and remainders three times per tile)
    auto dw_tile = [&](long long step, int &b, int &y0, int &x0) -> bool {   // patch of step `step` of this workgroup's walk
        if (step >= nsteps) return false;
        const long long tl = by_xcd ? (step & 7) * per_xcd + (step >> 3) : step;
        if (tl >= ntiles) return false;
        const unsigned tu = (unsigned)tl, pp = (unsigned)(ntx * nty);
        const unsigned bb = tu / pp, rem = tu - bb * pp, ty = rem / (unsigned)ntx;
        b = (int)bb;
        y0 = (int)ty * 4; x0 = (int)(rem - ty * (unsigned)ntx) * 16;
        return true;
    };
    auto dw_fetch = [&](int b, int y0, int x0) {   // thread t: f4 (cell, quad) = t, t + 256 of the (4 + 2) x (16 + 2) halo, per slice
        constexpr int C4 = NK0 * 4;
        const f4 *__restrict__ img = reinterpret_cast<const f4 *>(a.in) + (size_t)b * a.dw_H * a.dw_W * C4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = t + u * RC_THREADS;
            const int cell = i >> 2, quad = i & 3;
            const int gy = y0 - 1 + cell / 18, gx = x0 - 1 + cell % 18;
            const bool in = i < 6 * 18 * 4 && gy >= 0 && gy < a.dw_H && gx >= 0 && gx < a.dw_W;
            const f4 *__restrict__ src = img + ((size_t)(in ? gy : 0) * a.dw_W + (in ? gx : 0)) * C4 + quad;
#pragma unroll
            for (int kb = 0; kb < (DW ? NK0 : 1); ++kb) hva[kb][u] = in ? src[kb * 4] : f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    int cb = 0, cy0 = 0, cx0 = 0;   // DW: the patch whose halo is in hva
    if constexpr (DW) {
        if (dw_tile(blockIdx.x, cb, cy0, cx0)) dw_fetch(cb, cy0, cx0);
    }
    for (long long step = blockIdx.x; step < nsteps; step += gridDim.x) {
        const long long tl = by_xcd ? (step & 7) * per_xcd + (step >> 3) : step;
        if (tl >= ntiles) continue;   // (uniform over the workgroup)
        const long long tile0 = tl * 64;
        // (the weight pointers pass through an empty asm so the ~140 chunk addresses are formed inside the loop with
        //  scalar adds instead of being hoisted out of it as loop invariants, where they would take every register)
        asm volatile("" : "+s"(w1), "+s"(w2), "+s"(w3));
        // this lane's input row: channels [16 kb + 4 g, +4) of row tile0 + 16 wave + pos
        // (DW: the workgroup owns a 4 x 16 patch of cells of one image, wave = patch row, pos = cell in it)
        long long row = tile0 + 16 * wave + pos;
        bool live = row < a.rows;
        int dw_b = 0, dw_y0 = 0, dw_x0 = 0;
        if constexpr (DW) {
            dw_b = cb; dw_y0 = cy0; dw_x0 = cx0;
            live = dw_y0 + wave < a.dw_H && dw_x0 + pos < a.dw_W;
            row = ((long long)dw_b * a.dw_H + dw_y0 + wave) * a.dw_W + dw_x0 + pos;
        }
        if (!live) row = a.rows - 1;
        const long long out_row = row;
        f4 x0[NK0];
        if constexpr (!DW) {
            if (AHEAD && tl != (long long)blockIdx.x) {
#pragma unroll
                for (int kb = 0; kb < NK0; ++kb) x0[kb] = xn[kb];   // requested under the previous tile's second layer
            } else {
                const float *__restrict__ src = a.in + (size_t)row * a.in_stride + 4 * g;
#pragma unroll
                for (int kb = 0; kb < NK0; ++kb) x0[kb] = *reinterpret_cast<const f4 *>(src + 16 * kb);
            }
        } else {
            // depthwise 3x3 + shift + ReLU: the patch's halo is staged through LDS (each cell of the map is fetched once per
            // workgroup instead of up to nine times), from the registers filled a tile ahead, DW_SL slices of 16 channels per
            // stage.  Inside a stage the work is dealt differently from the chain: WAVE w takes channel quad w of a slice for
            // all 64 cells of the patch (lane = cell), so the nine tap weights and the shift of (slice, quad) are wave-uniform
            // and come through the scalar cache — no LDS reads and no registers for them (the first form read them from LDS per
            // lane, 9 of its 19 reads per slice, and the prologue was bound by the LDS pipe: 152 KB per wave and tile,
            // 11.3k of a tile's 27.2k cycles, tools/diag/bev_head_phase.py).  The results cross to the chain's fragment
            // layout (lane (pos, g) of wave r holds quad g of cell (r, pos)) through a padded LDS tile.
            // Same fma order per channel as before (shift, then taps row by row, left to right): identical results.
            constexpr int C4 = NK0 * 4;
            static_assert(NK0 % DW_SL == 0, "rows_chain DW: slices per stage");
            const int wu = __builtin_amdgcn_readfirstlane(wave);
            const int cr = lane >> 4;                                   // this lane's cell of the patch: row cr, column pos
#if RC_DIAG & 4
            long long dws[10];
            int dwi = 0;
#define DW_STAMP() do { __builtin_amdgcn_s_waitcnt(0xc07f); if (dwi < 9) dws[dwi++] = __builtin_amdgcn_s_memtime(); } while (0)
            { __builtin_amdgcn_s_waitcnt(0x0070 | 0xc00f); }   // (lgkm and vm counters left to the stamps below)
            dws[dwi++] = __builtin_amdgcn_s_memtime();
#else
#define DW_STAMP()
#endif
#pragma unroll
            for (int s0 = 0; s0 < NK0; s0 += DW_SL) {
#pragma unroll
                for (int q = 0; q < DW_SL; ++q) {
                    // quad-major inside a slice ((quad, cell) instead of (cell, quad)): a wave reads ONE quad of 64 cells per
                    // tap, consecutive lanes 16 bytes apart (cell-major, those reads used a quarter of the banks: 497 us against 448)
                    if (t < 6 * 18 * 4) halo[q * (6 * 18 * 4) + (t & 3) * (6 * 18) + (t >> 2)] = hva[s0 + q][0];
                    if (t + RC_THREADS < 6 * 18 * 4) halo[q * (6 * 18 * 4) + (t & 3) * (6 * 18) + ((t + RC_THREADS) >> 2)] = hva[s0 + q][1];
                }
                DW_STAMP();
                __syncthreads();   // halo of this stage written; everyone has read the stage before's crossing tile
                DW_STAMP();
#pragma unroll
                for (int q = 0; q < DW_SL; ++q) {
                    const int kb = s0 + q;
                    // (constant address space + a wave-uniform address = s_load_dwordx4; through a plain pointer the compiler
                    //  issues per-lane global loads, since the kernel's own stores might alias)
                    const cf4c wq = (cf4c)(uintptr_t)(a.dw_w + kb * 16 + 4 * wu);
                    f4 acc = *(cf4c)(uintptr_t)(a.dw_shift + kb * 16 + 4 * wu);
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const f4 v = halo[q * (6 * 18 * 4) + wu * (6 * 18) + (cr + dy) * 18 + pos + dx];
                            const f4 k = wq[(dy * 3 + dx) * C4];
                            acc = __builtin_elementwise_fma(k, v, acc);   // two v_pk_fma_f32: the same IEEE fma per channel
                        }
                    acc = f4{fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f)};
                    xs[(q * 64 + lane) * 5 + wu] = acc;
                    if ((q % DW_PIN) == DW_PIN - 1) {
                        asm volatile("" ::: "memory");   // the slice's taps are consumed here, before the next slice's reads are issued
                        __builtin_amdgcn_sched_barrier(0);   // (unpinned, the compiler gathers every tap read up front and spills them)
                    }
                }
                DW_STAMP();
                __syncthreads();   // crossing tile written; every tap of this stage read (the next stage may overwrite the halo)
                DW_STAMP();
#pragma unroll
                for (int q = 0; q < DW_SL; ++q) x0[s0 + q] = xs[(q * 64 + wave * 16 + pos) * 5 + g];
            }
#if RC_DIAG & 4
            {
                const long long k = (step - blockIdx.x) / gridDim.x;
                if (t == 0 && k == 1)
                    for (int i = 0; i < 9; ++i) reinterpret_cast<long long *>(a.out)[(size_t)gridDim.x * 8 + (size_t)blockIdx.x * 9 + i] = dws[i];
            }
#endif
            if (dw_tile(step + gridDim.x, cb, cy0, cx0)) dw_fetch(cb, cy0, cx0);   // the next tile's halo
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
#if RC_DIAG & 4   // phase timestamps of wave 0 (first two tiles of each workgroup) instead of results: tools/diag/rows_chain_phase.py
        long long ts[4];
        ts[0] = __builtin_amdgcn_s_memtime();
#endif
        f4 x1[NK1];
        rc_layer<NK0, NK1>(x0, x1, w1, a.bias + a.boff[0], lds, p, (NL > 1 || a.relu_last) ? 0.0f : neg_inf, NL > 1 ? w2 : w1,
                           NL > 1 ? NK1 : NK0, NL > 1 ? NK2 : NK1, t, lane, r, an);
        if constexpr (NL == 1) {
            store(x1, NK1);
        } else {
            f4 x2[NK2 ? NK2 : 1];
            if constexpr (AHEAD) {   // the next tile's input rows: in flight under this tile's second (largest) layer
                long long nrow = (tl + gridDim.x) * 64 + 16 * wave + pos;
                if (nrow >= a.rows) nrow = a.rows - 1;
                const float *__restrict__ nsrc = a.in + (size_t)nrow * a.in_stride + 4 * g;
#pragma unroll
                for (int kb = 0; kb < NK0; ++kb)
#if RC_DIAG & 32   // timing build: the next tile's rows are not read
                    xn[kb] = f4{(float)nrow, 1.f, 2.f, 3.f};
#else
                    xn[kb] = *reinterpret_cast<const f4 *>(nsrc + 16 * kb);
#endif
            }
#if RC_DIAG & 4
            asm volatile("" :: "v"(x1[0].x));
            ts[1] = __builtin_amdgcn_s_memtime();
#endif
            rc_layer<NK1, (NK2 ? NK2 : 1)>(x1, x2, w2, a.bias + a.boff[1], lds, p, (NL > 2 || a.relu_last) ? 0.0f : neg_inf,
                                           NL > 2 ? w3 : w1, NL > 2 ? rc_next_nkb(NK2, NK3) : NK0, NL > 2 ? NK3 : NK1, t, lane, r, an);
            if constexpr (NL == 2) {
                store(x2, NK2);
            } else {
                f4 x3[NK3 ? NK3 : 1];
#if RC_DIAG & 4
                asm volatile("" :: "v"(x2[0].x));
                ts[2] = __builtin_amdgcn_s_memtime();
#endif
                rc_layer<(NK2 ? NK2 : 4), (NK3 ? NK3 : 1)>(x2, x3, w3, a.bias + a.boff[2], lds, p, a.relu_last ? 0.0f : neg_inf, w1, NK0, NK1, t,
                                                          lane, r, an);
#if RC_DIAG & 4
                asm volatile("" :: "v"(x3[0].x));
                ts[3] = __builtin_amdgcn_s_memtime();
                const long long k = (step - blockIdx.x) / gridDim.x;   // (== (tl - blockIdx.x) / gridDim.x in launch order)
                if (t == 0 && k < 2)
                    for (int i = 0; i < 4; ++i) reinterpret_cast<long long *>(a.out)[((size_t)blockIdx.x * 2 + k) * 4 + i] = ts[i];
#else
                store(x3, NK3);
#endif
            }
        }
    }
}

// ---- two chains over the SAME rows in one launch (the point head's class and box branches) --------------------------------
// /root/reference/pcdet/models/dense_heads/point_head_box.py:7-60 is one module with two make_fc_layers stacks on one input:
// here a wave keeps its tile's input fragments in registers, runs branch A's three layers, then branch B's, and the weight
// chunks of the six layers form ONE continuous stream (A1 A2 A3 B1 B2 B3 A1 ...).  Against two launches of rows_chain_kernel:
// the rows are read once (268 MB per launch at the bench shape), a workgroup's slow first tile is paid once, one launch
// boundary less.  Every layer is rc_layer as above, so both outputs are bit-identical to the two-launch form.
struct RowsChainPairArgs {
    int rows, in_stride;
    const float *in;
    const float *wpack[2], *bias[2];   // branch A, branch B: the same widths (dims), packed as for rows_chain_kernel
    int woff[3], boff[3];
    float *out[2];
    int out_stride[2], cout[2], relu_last;
};

template <int NK0, int NK1, int NK2, int NK3>
__global__ __launch_bounds__(RC_THREADS, 2) void rows_chain_pair_kernel(RowsChainPairArgs a) {
    static_assert(NK1 > 0 && NK2 > 0 && NK3 > 0, "rows_chain_pair: three layers");
    __shared__ __attribute__((aligned(16))) f4 lds[2 * RC_CHUNK_F4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int pos = lane & 15, g = lane >> 4;
    const f4 *wa1 = reinterpret_cast<const f4 *>(a.wpack[0] + a.woff[0]), *wa2 = reinterpret_cast<const f4 *>(a.wpack[0] + a.woff[1]),
             *wa3 = reinterpret_cast<const f4 *>(a.wpack[0] + a.woff[2]);
    const f4 *wb1 = reinterpret_cast<const f4 *>(a.wpack[1] + a.woff[0]), *wb2 = reinterpret_cast<const f4 *>(a.wpack[1] + a.woff[1]),
             *wb3 = reinterpret_cast<const f4 *>(a.wpack[1] + a.woff[2]);
    const float neg_inf = -__builtin_inff();
    const float last_floor = a.relu_last ? 0.0f : neg_inf;
    f4 r[4];
    const long long ntiles = ((long long)a.rows + 63) / 64;
    int p = 0;
    RC_FETCH(r, lds, wa1, NK0, 0, 0, NK1 < 4 ? NK1 : 4, t);
    RC_STASH(r, lds, NK1 < 4 ? NK1 : 4, t);
    __syncthreads();
    f4 an[4];
    rc_first_fragments(an, lds, 0, lane);
    f4 xn[NK0];
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        // (weight pointers through an empty asm: see rows_chain_kernel)
        asm volatile("" : "+s"(wa1), "+s"(wa2), "+s"(wa3), "+s"(wb1), "+s"(wb2), "+s"(wb3));
        long long row = tl * 64 + 16 * wave + pos;
        const bool live = row < a.rows;
        if (!live) row = a.rows - 1;
        f4 x0[NK0];
        if (tl != (long long)blockIdx.x) {
#pragma unroll
            for (int kb = 0; kb < NK0; ++kb) x0[kb] = xn[kb];   // requested under the previous tile's branch B
        } else {
            const float *__restrict__ src = a.in + (size_t)row * a.in_stride + 4 * g;
#pragma unroll
            for (int kb = 0; kb < NK0; ++kb) x0[kb] = *reinterpret_cast<const f4 *>(src + 16 * kb);
        }
        auto store = [&](int br, auto &y) {
            if (!live) return;
            float *__restrict__ orow = a.out[br] + (size_t)row * a.out_stride[br];
            const int cout = a.cout[br];
#pragma unroll
            for (int mb = 0; mb < NK3; ++mb) {
                const int c0 = 16 * mb + 4 * g;
                if (c0 + 4 <= cout) *reinterpret_cast<f4 *>(orow + c0) = y[mb];
                else {
                    if (c0 < cout) orow[c0] = y[mb].x;
                    if (c0 + 1 < cout) orow[c0 + 1] = y[mb].y;
                    if (c0 + 2 < cout) orow[c0 + 2] = y[mb].z;
                }
            }
        };
        {   // branch A (x0 stays live for branch B: it takes the place the next tile's rows hold in rows_chain_kernel)
            f4 x1[NK1], x2[NK2], x3[NK3];
            rc_layer<NK0, NK1>(x0, x1, wa1, a.bias[0] + a.boff[0], lds, p, 0.0f, wa2, NK1, NK2, t, lane, r, an);
            rc_layer<NK1, NK2>(x1, x2, wa2, a.bias[0] + a.boff[1], lds, p, 0.0f, wa3, rc_next_nkb(NK2, NK3), NK3, t, lane, r, an);
            rc_layer<NK2, NK3>(x2, x3, wa3, a.bias[0] + a.boff[2], lds, p, last_floor, wb1, NK0, NK1, t, lane, r, an);
            store(0, x3);
        }
        {   // branch B
            f4 x1[NK1], x2[NK2], x3[NK3];
            rc_layer<NK0, NK1>(x0, x1, wb1, a.bias[1] + a.boff[0], lds, p, 0.0f, wb2, NK1, NK2, t, lane, r, an);
            {   // the next tile's input rows: in flight under this branch's second (largest) layer
                long long nrow = (tl + gridDim.x) * 64 + 16 * wave + pos;
                if (nrow >= a.rows) nrow = a.rows - 1;
                const float *__restrict__ nsrc = a.in + (size_t)nrow * a.in_stride + 4 * g;
#pragma unroll
                for (int kb = 0; kb < NK0; ++kb) xn[kb] = *reinterpret_cast<const f4 *>(nsrc + 16 * kb);
            }
            rc_layer<NK1, NK2>(x1, x2, wb2, a.bias[1] + a.boff[1], lds, p, 0.0f, wb3, rc_next_nkb(NK2, NK3), NK3, t, lane, r, an);
            rc_layer<NK2, NK3>(x2, x3, wb3, a.bias[1] + a.boff[2], lds, p, last_floor, wa1, NK0, NK1, t, lane, r, an);
            store(1, x3);
        }
    }
}

// ---- FP module in the hoisted form, chain in registers ---------------------------------------------------------------
//   h1[r] = relu(sum_k w_k z[idx_k[r]] + W1s skip[r] + b1),   out[r] = relu(W2 h1[r] + b2)
// (fused_mlp.hip "pre" form: z = W1[:, known] f was made over the m known points of each cloud.)  Same ownership as
// rows_chain_kernel: a wave holds 16 rows and every channel; the three z rows of a point arrive as the lane's own
// float4s of the accumulator layout (channels 16 mb + 4 g + i of row pos), requested before the skip GEMM so they
// land under its MFMAs.  NK0 = k-blocks of the skip input (0: c_skip <= 4, applied with scalar FMAs), NK1 / NK2 =
// blocks of h1 / out.
#ifndef FPC_DIAG
#define FPC_DIAG 0
#endif
#ifndef FPC_LATE
#define FPC_LATE 1   // with a skip GEMM, request the z rows behind it (0: under it — 23 spilled registers at FP2's widths, 246 us against 237)
#endif
struct FpChainArgs {
    int nt_out;                 // 1: the output rows leave through non-temporal stores (they are read much later, by another
                                // kernel; kept out of the L2 they no longer displace the z rows the gathers come back for)
    int rows, n, m, c_skip, z_stride;
    const float *z, *skip, *weight;
    const int *idx;
    const float *wpack, *bias;
    int woff[2], boff[2];
    float *out;
    int out_stride, cout;
    // HEAD instantiations: the point head's two stacks (RowsChainPairArgs' fields) run on the tile's output rows while they
    // are still in registers
    const float *hw[2], *hb[2];
    int hwoff[3], hboff[3];
    float *hout[2];
    int hout_stride[2], hcout[2], hrelu_last;
};

// HEAD: the last FP module and the point head in ONE launch.  /root/reference/pcdet/models/backbones_3d/pointnet2_backbone.py:
// 96-111 ends with FP module 1 writing point_features, which dense_heads/point_head_box.py:71-76 reads straight back as the
// input of its two stacks.  The module's output fragments (lane (pos, g): channels 16 mb + 4 g .. of row pos) ARE the
// stacks' input fragments, so after its rows have left for memory (point_features is an output of the detector) the wave
// runs class stack, then box stack on them as rows_chain_pair_kernel does.  The FP module's gathers and its 268 MB of
// output stores sit in a weight stream of seven layers instead of one (alone they add up with its MFMAs: DESIGN 7h), the
// head does not read the rows back, one launch boundary and one slow first tile per workgroup less.  Every layer is
// rc_layer: the FP output equals fp_chain_kernel<0, 8, 8>'s and the logits equal rows_chain_pair_kernel's on it, bit for bit.
template <int NK0, int NK1, int NK2, bool HEAD = false>
__global__ __launch_bounds__(RC_THREADS, 2) void fp_chain_kernel(FpChainArgs a) {
    static_assert(NK1 % 8 == 0 && NK2 % 8 == 0, "fp_chain: widths are multiples of 128 channels (one staging pass)");
    static_assert(!HEAD || (NK0 == 0 && NK2 == 8), "fp_chain HEAD: FP module 1 (raw skip channels, 128 out) + the 128 -> 256 -> 256 -> <= 16 stacks");
    constexpr int H1 = 16, H2 = 16, H3 = 1;   // HEAD: blocks of the stacks' layers
    __shared__ __attribute__((aligned(16))) f4 lds[2 * RC_CHUNK_F4];
    // per-wave transposition buffer: 16 rows x 128 channels (+1 quad per row: D-layout reads of a 16-lane group then
    // touch 16 different bank quads).  Gathers and stores move whole 128-byte lines per row (8 lanes x 16 B); the
    // accumulator layout (lane (pos, g) = channels 16 mb + 4 g + i of row pos) would move 64-byte halves of them.
    constexpr int SQ = 33;
    __shared__ __attribute__((aligned(16))) f4 stage_all[4 * 16 * SQ];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int pos = lane & 15, g = lane >> 4;
    const int grow = lane >> 3, gq = lane & 7;   // line-wise role: rows grow and grow + 8 of the wave's tile, quad gq + 8 j
    f4 *stage = stage_all + wave * 16 * SQ;
    const f4 *w1 = reinterpret_cast<const f4 *>(a.wpack + a.woff[0]);
    const f4 *w2 = reinterpret_cast<const f4 *>(a.wpack + a.woff[1]);
    const f4 *wa1 = HEAD ? reinterpret_cast<const f4 *>(a.hw[0] + a.hwoff[0]) : nullptr, *wa2 = HEAD ? reinterpret_cast<const f4 *>(a.hw[0] + a.hwoff[1]) : nullptr,
             *wa3 = HEAD ? reinterpret_cast<const f4 *>(a.hw[0] + a.hwoff[2]) : nullptr;
    const f4 *wb1 = HEAD ? reinterpret_cast<const f4 *>(a.hw[1] + a.hwoff[0]) : nullptr, *wb2 = HEAD ? reinterpret_cast<const f4 *>(a.hw[1] + a.hwoff[1]) : nullptr,
             *wb3 = HEAD ? reinterpret_cast<const f4 *>(a.hw[1] + a.hwoff[2]) : nullptr;
    const float neg_inf = -__builtin_inff();
    f4 r[4];
    const long long ntiles = ((long long)a.rows + 63) / 64;
    // Workgroups are dealt to the 8 XCDs round-robin; when the clouds split evenly, XCD x walks clouds x, x + 8, ... so
    // its L2 holds one cloud's z rows at a time (2 MB at FP1's shape) instead of a slice of every cloud in flight.
    const int tpc = a.n / 64, nclouds = a.rows / a.n;
    const bool by_xcd = tpc * 64 == a.n && (nclouds & 7) == 0 && (gridDim.x & 7) == 0;
    const long long per_xcd = by_xcd ? (long long)(nclouds >> 3) * tpc : 0;
    const long long first = by_xcd ? (blockIdx.x >> 3) : blockIdx.x, step = by_xcd ? (gridDim.x >> 3) : gridDim.x,
                    last = by_xcd ? per_xcd : ntiles;
    const float *__restrict__ zbase = a.z;
    auto tile_of = [&](long long u) { return by_xcd ? ((u / tpc) * 8 + (blockIdx.x & 7)) * tpc + u % tpc : u; };
    // line-wise rows of this lane in tile tl: neighbours as 32-bit element offsets from the uniform base
    // (b m z_stride < 2^31, host check), their weights
    auto neighbours = [&](long long tl, unsigned (&ze)[2][3], float (&zw)[2][3]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            long long rr = tl * 64 + 16 * wave + grow + 8 * h;
            if (rr >= a.rows) rr = a.rows - 1;
            const unsigned zrow0 = (unsigned)(rr / a.n) * (unsigned)a.m;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ze[h][k] = (zrow0 + (unsigned)a.idx[rr * 3 + k]) * (unsigned)a.z_stride + 4u * gq;
                zw[h][k] = a.weight[rr * 3 + k];
            }
        }
    };
    unsigned ze[2][3];
    float zw[2][3];
    f4 zv[2][3][4];
    auto request = [&](int cb) {   // channels [cb, cb + 128) of the six rows
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#if FPC_DIAG & 1   // timing build (tools/diag/fp_chain_rate.py): no z gathers
                    zv[h][k][j] = f4{zw[h][k], 0.f, 0.f, 0.f};
                    continue;
#endif
                    zv[h][k][j] = *reinterpret_cast<const f4 *>(zbase + (ze[h][k] + (unsigned)(cb + 32 * j)));
                }
    };
    // interpolate (pinned order, as three_interpolate) and park the 16 x 128 block in the staging buffer
    auto park = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f4 z0 = zv[h][0][j], z1 = zv[h][1][j], z2 = zv[h][2][j];
                f4 v;
                v.x = fmaf(zw[h][2], z2.x, fmaf(zw[h][1], z1.x, zw[h][0] * z0.x));
                v.y = fmaf(zw[h][2], z2.y, fmaf(zw[h][1], z1.y, zw[h][0] * z0.y));
                v.z = fmaf(zw[h][2], z2.z, fmaf(zw[h][1], z1.z, zw[h][0] * z0.z));
                v.w = fmaf(zw[h][2], z2.w, fmaf(zw[h][1], z1.w, zw[h][0] * z0.w));
                stage[(grow + 8 * h) * SQ + gq + 8 * j] = v;
            }
    };
    // The weight chunks stream without a stop at tile boundaries: the first chunk of a tile is prefetched behind the
    // last chunk of the tile before.  The z rows of a tile are requested one tile ahead (see the loop).
    int p = 0;
    if constexpr (NK0 > 0) {
        RC_FETCH(r, lds, w1, NK0, 0, 0, NK1 < 4 ? NK1 : 4, t, NK0 < 4 ? NK0 : 4);
        RC_STASH(r, lds, NK1 < 4 ? NK1 : 4, t);
    } else {
        RC_FETCH(r, lds, w2, NK1, 0, 0, NK2 < 4 ? NK2 : 4, t, NK1 < 4 ? NK1 : 4);
        RC_STASH(r, lds, NK2 < 4 ? NK2 : 4, t);
    }
    __syncthreads();
    f4 an[4];
    rc_first_fragments(an, lds, 0, lane);
    // A workgroup takes CONSECUTIVE tiles (u = first * per_wg + k).  With u = first + k * step the workgroups resident on an XCD at one
    // time (64 of its 384) moved on to tiles `step` further along after their first one — a different cloud — so one L2 held the z
    // rows of up to three clouds (6 MB against its 4 MB) and every tile's 192 scattered row gathers missed: FETCH 292 MB per FP1
    // launch for 80 MB of distinct rows (PMC).
    const long long per_wg = (last + step - 1) / step;
    for (long long k = 0; k < per_wg; ++k) {
        const long long u = first * per_wg + k;
        if (u >= last) break;
        const long long tl = tile_of(u);
        asm volatile("" : "+s"(w1), "+s"(w2));
        if constexpr (HEAD) asm volatile("" : "+s"(wa1), "+s"(wa2), "+s"(wa3), "+s"(wb1), "+s"(wb2), "+s"(wb3));
        const long long wrow0 = tl * 64 + 16 * wave;
        long long row = wrow0 + pos;          // accumulator-layout row of this lane
        if (row >= a.rows) row = a.rows - 1;
        const long long lrow[2] = {wrow0 + grow, wrow0 + grow + 8};
        neighbours(tl, ze, zw);
        if constexpr (NK0 == 0 || !FPC_LATE) request(0);
        f4 x1[NK1];
        if constexpr (NK0 > 0) {
            f4 x0[NK0];
            const float *__restrict__ src = a.skip + (size_t)row * a.c_skip + 4 * g;
#pragma unroll
            for (int kb = 0; kb < NK0; ++kb) {
                x0[kb] = f4{0.f, 0.f, 0.f, 0.f};
                if (16 * kb + 4 * g + 4 <= a.c_skip) x0[kb] = *reinterpret_cast<const f4 *>(src + 16 * kb);   // c_skip % 4 == 0 (host check)
            }
            rc_layer<NK0, NK1>(x0, x1, w1, a.bias + a.boff[0], lds, p, neg_inf, w2, NK1, NK2, t, lane, r, an);
            if constexpr (FPC_LATE) request(0);
        } else {
            // c_skip <= 4 raw input channels; the packed layer-1 weights are zero past c_skip, so all four terms are formed
            const float *__restrict__ sp = a.skip + (size_t)row * a.c_skip;
            const float s0 = a.c_skip > 0 ? sp[0] : 0.f, s1 = a.c_skip > 1 ? sp[1] : 0.f, s2 = a.c_skip > 2 ? sp[2] : 0.f,
                        s3 = a.c_skip > 3 ? sp[3] : 0.f;
#pragma unroll
            for (int mb = 0; mb < NK1; ++mb) {
                x1[mb] = *reinterpret_cast<const f4 *>(a.bias + a.boff[0] + 16 * mb + 4 * g);
                gf4c wr = (gf4c)(w1 + (size_t)mb * 64 + 4 * g);   // layer 1 packed with ONE k-block: W1[16 mb + 4 g + j][0..3] = wr[j]
                const f4 q0 = wr[0], q1 = wr[1], q2 = wr[2], q3 = wr[3];
                x1[mb].x += fmaf(q0.w, s3, fmaf(q0.z, s2, fmaf(q0.y, s1, q0.x * s0)));
                x1[mb].y += fmaf(q1.w, s3, fmaf(q1.z, s2, fmaf(q1.y, s1, q1.x * s0)));
                x1[mb].z += fmaf(q2.w, s3, fmaf(q2.z, s2, fmaf(q2.y, s1, q2.x * s0)));
                x1[mb].w += fmaf(q3.w, s3, fmaf(q3.z, s2, fmaf(q3.y, s1, q3.x * s0)));
            }
        }
        // h1 = relu(. + interpolated z), 128 channels per staging pass
#pragma unroll
        for (int mb0 = 0; mb0 < NK1; mb0 += 8) {
            park();
            if (mb0 + 8 < NK1) request(16 * (mb0 + 8));
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f4 v = stage[pos * SQ + 4 * i + g];
                f4 &h = x1[mb0 + i];
                h.x = fmaxf(h.x + v.x, 0.f); h.y = fmaxf(h.y + v.y, 0.f); h.z = fmaxf(h.z + v.z, 0.f); h.w = fmaxf(h.w + v.w, 0.f);
            }
            __builtin_amdgcn_wave_barrier();
        }
        f4 x2[NK2];
        // one line-wise piece (row grow + 8 h, quads gq + 8 j of channel block mb0) from the staging buffer to memory
        auto store_piece = [&](const long long (&lr)[2], int mb0, int h, int j) {
            if (lr[h] >= a.rows) return;
            float *__restrict__ orow = a.out + (size_t)lr[h] * a.out_stride + 16 * mb0;
            const f4 v = stage[(grow + 8 * h) * SQ + gq + 8 * j];
            const int c0 = 16 * mb0 + 4 * (gq + 8 * j);
            if (c0 + 4 <= a.cout) {
                if (a.nt_out) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(orow + 4 * (gq + 8 * j)));
                else *reinterpret_cast<f4 *>(orow + 4 * (gq + 8 * j)) = v;
            } else {
                if (c0 < a.cout) orow[4 * (gq + 8 * j)] = v.x;
                if (c0 + 1 < a.cout) orow[4 * (gq + 8 * j) + 1] = v.y;
                if (c0 + 2 < a.cout) orow[4 * (gq + 8 * j) + 2] = v.z;
            }
        };
#if FPC_DIAG & 4   // timing build: no second layer
#pragma unroll
        for (int mb = 0; mb < NK2; ++mb) x2[mb] = x1[mb % NK1];
        __syncthreads();
#else
        if constexpr (HEAD) rc_layer<NK1, NK2>(x1, x2, w2, a.bias + a.boff[1], lds, p, 0.0f, wa1, NK2, H1, t, lane, r, an);
        else rc_layer<NK1, NK2>(x1, x2, w2, a.bias + a.boff[1], lds, p, 0.0f, NK0 > 0 ? w1 : w2, NK0 > 0 ? NK0 : NK1, NK0 > 0 ? NK1 : NK2, t, lane, r, an);
#endif
        // rows leave line-wise through the same buffer.  (Measured and dropped: the next tile's z rows requested ahead
        // of these stores, 272 us against 258 at FP1's shape; the stores deferred by a tile and spread behind the next
        // tile's weight chunks, 407 us — vmcnt retires in order, so every chunk's prefetch then waits for stores.)
#if !(FPC_DIAG & 2)
#pragma unroll
        for (int mb0 = 0; mb0 < NK2; mb0 += 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) stage[pos * SQ + 4 * i + g] = x2[mb0 + i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) store_piece(lrow, mb0, h, j);
            __builtin_amdgcn_wave_barrier();
        }
#endif
        if constexpr (HEAD) {
            const bool live = wrow0 + pos < a.rows;
            const float last_floor = a.hrelu_last ? 0.0f : neg_inf;
            auto hstore = [&](int br, auto &y) {
                if (!live) return;
                float *__restrict__ orow = a.hout[br] + (size_t)row * a.hout_stride[br];
                const int cout = a.hcout[br];
#pragma unroll
                for (int mb = 0; mb < H3; ++mb) {
                    const int c0 = 16 * mb + 4 * g;
                    if (c0 + 4 <= cout) *reinterpret_cast<f4 *>(orow + c0) = y[mb];
                    else {
                        if (c0 < cout) orow[c0] = y[mb].x;
                        if (c0 + 1 < cout) orow[c0 + 1] = y[mb].y;
                        if (c0 + 2 < cout) orow[c0 + 2] = y[mb].z;
                    }
                }
            };
            {   // class stack (x2 stays live for the box stack)
                f4 y1[H1], y2[H2], y3[H3];
                rc_layer<NK2, H1>(x2, y1, wa1, a.hb[0] + a.hboff[0], lds, p, 0.0f, wa2, H1, H2, t, lane, r, an);
                rc_layer<H1, H2>(y1, y2, wa2, a.hb[0] + a.hboff[1], lds, p, 0.0f, wa3, rc_next_nkb(H2, H3), H3, t, lane, r, an);
                rc_layer<H2, H3>(y2, y3, wa3, a.hb[0] + a.hboff[2], lds, p, last_floor, wb1, NK2, H1, t, lane, r, an);
                hstore(0, y3);
            }
            {   // box stack; behind its last chunk: the first chunk of the FP module's layer for the next tile
                f4 y1[H1], y2[H2], y3[H3];
                rc_layer<NK2, H1>(x2, y1, wb1, a.hb[1] + a.hboff[0], lds, p, 0.0f, wb2, H1, H2, t, lane, r, an);
                rc_layer<H1, H2>(y1, y2, wb2, a.hb[1] + a.hboff[1], lds, p, 0.0f, wb3, rc_next_nkb(H2, H3), H3, t, lane, r, an);
                rc_layer<H2, H3>(y2, y3, wb3, a.hb[1] + a.hboff[2], lds, p, last_floor, w2, NK1, NK2, t, lane, r, an);
                hstore(1, y3);
            }
        }
    }
}

static int g_rc_dw_xcd = 1;       // heat-map kernel: patches dealt to the XCDs in contiguous ranges (0: launch order)
extern "C" int pdm_tune_rows_chain_xcd(int on) { const int old = g_rc_dw_xcd; g_rc_dw_xcd = on != 0; return old; }
static int g_rc_wg_per_cu = 12;   // grid cap of the chain kernels = 256 CUs x this many workgroups (2 are resident at a time)
extern "C" int pdm_tune_rows_chain_wg_per_cu(int n) { const int old = g_rc_wg_per_cu; if (n > 0) g_rc_wg_per_cu = n; return old; }
// the same for the heat-map head's one-kernel form: its workgroups pay more at their start (first halo and first weight chunk
// not prefetched), so fewer, longer ones: alone 436 / 442 / 452 us at 2 / 4 / 12 (tools/diag/bev_head_rate.py), in the step 4.37 / 4.40 / 4.41 ms
static int g_rc_dw_wg_per_cu = 2;
extern "C" int pdm_tune_rows_chain_dw_wg_per_cu(int n) { const int old = g_rc_dw_wg_per_cu; if (n > 0) g_rc_dw_wg_per_cu = n; return old; }
static int g_fpc_nt = 0;        // FP chain kernel: non-temporal output stores (off: no change in time or in PMC traffic; the rows are the point head's input next)
extern "C" int pdm_tune_fp_chain_nt(int on) { const int old = g_fpc_nt; g_fpc_nt = on != 0; return old; }
// which FP shapes take the chain kernel: bit 0 the FP1 shape, bit 1 the FP2 shape (else the LDS-tiled forms).  Default 2: with the
// backbone's REAL three-NN indices (spatially local gathers) the LDS-tiled kernel runs FP1 in 243 us against 252-268 us for the
// chain (tools/diag/fp_chain_rate.py --real; round 2 chose the chain on random indices: 258 against 275) and moves ~410 MB per
// launch against 449-570; FP2 stays on the chain (209 against 276 us).  The pipelined step is the same either way (4.45-4.50 ms).
static int g_fpc_mask = 2;
extern "C" int pdm_tune_fp_chain_mask(int m) { const int old = g_fpc_mask; if (m >= 0) g_fpc_mask = m & 3; return old; }
static int g_fpc_pad_lds = 0;   // diagnostic: extra dynamic LDS per workgroup (forces one workgroup per CU at 90 KB)
extern "C" int pdm_tune_fp_chain_pad_lds(int bytes) { const int old = g_fpc_pad_lds; g_fpc_pad_lds = bytes; return old; }

// Returns 1 in *launched when an instantiation fits (two-layer FP module, hoisted form, many rows).
int fp_chain_launch(void *stream, int b, int n, int m, int c_skip, const float *z_pm, int z_stride, const float *skip_pm,
                    const int *idx, const float *weight, const int *dims, const float *wpack, const float *bias, float *out_pm,
                    int out_stride, int cout, int *launched) {
    *launched = 0;
    const long long rows = (long long)b * n;
    if (rows < 32768 || rows >= (1ll << 31) || (long long)b * m * z_stride >= (1ll << 31)) return 0;
    FpChainArgs a{};
    a.nt_out = g_fpc_nt;
    a.rows = (int)rows; a.n = n; a.m = m; a.c_skip = c_skip; a.z_stride = z_stride;
    a.z = z_pm; a.skip = skip_pm; a.weight = weight; a.idx = idx; a.wpack = wpack; a.bias = bias;
    a.woff[0] = 0; a.boff[0] = 0; a.woff[1] = dims[0] * dims[1]; a.boff[1] = dims[1];
    a.out = out_pm; a.out_stride = out_stride; a.cout = cout;
    const long long tiles = (rows + 63) / 64;
    int grid = (int)(tiles < 256 * g_rc_wg_per_cu ? tiles : 256 * g_rc_wg_per_cu);
    if (grid >= 8) grid &= ~7;   // a multiple of the 8 XCDs (the tile loop covers the rest)
#define FC_TRY(K0, K1, K2, COND)                                                                                      \
    if ((COND) && dims[1] == 16 * K1 && dims[2] == 16 * K2) {                                                        \
        if (g_fpc_pad_lds > 65536 - 50000) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fp_chain_kernel<K0, K1, K2>), hipFuncAttributeMaxDynamicSharedMemorySize, g_fpc_pad_lds); /* a tuning knob: set on every call */ \
        hipLaunchKernelGGL((fp_chain_kernel<K0, K1, K2>), dim3(grid), dim3(RC_THREADS), g_fpc_pad_lds, as_stream(stream), a);    \
        *launched = 1;                                                                                               \
        return check_launch("fp_mlp_fused_pre(chain)");                                                              \
    }
    FC_TRY(0, 8, 8, (g_fpc_mask & 1) && c_skip <= 4 && dims[0] == 16)                               // FP1: raw input channels -> 128 -> 128
    FC_TRY(6, 16, 16, (g_fpc_mask & 2) && c_skip % 4 == 0 && c_skip > 4 && dims[0] == 96)           // FP2: 96 skip channels -> 256 -> 256
#undef FC_TRY
    return 0;
}

static int g_fph_tiles = 2;   // pdm_fp_head_fused: tiles per workgroup
}  // namespace pdm
extern "C" int pdm_tune_fp_head_tiles(int n) { const int old = pdm::g_fph_tiles; if (n > 0) pdm::g_fph_tiles = n; return old; }
namespace pdm {
// FP module (hoisted form, raw skip channels, 16 -> 128 -> 128) + the point head's two 128 -> 256 -> 256 -> <= 16 stacks in one
// launch (fp_chain_kernel<0, 8, 8, true>).  Returns 1 in *launched when the shapes fit.
int fp_head_launch(void *stream, int b, int n, int m, int c_skip, const float *z_pm, int z_stride, const float *skip_pm,
                   const int *idx, const float *weight, const int *dims, const float *wpack, const float *bias, float *out_pm,
                   int out_stride, int cout, const int *hdims, const float *hw_a, const float *hb_a, const float *hw_b,
                   const float *hb_b, int relu_last, float *out_a, int out_stride_a, int cout_a, float *out_b, int out_stride_b,
                   int cout_b, int *launched) {
    *launched = 0;
    const long long rows = (long long)b * n;
    if (rows < 32768 || rows >= (1ll << 31) || (long long)b * m * z_stride >= (1ll << 31)) return 0;
    if (!(c_skip <= 4 && dims[0] == 16 && dims[1] == 128 && dims[2] == 128 && hdims[0] == 128 && hdims[1] == 256 && hdims[2] == 256 &&
          hdims[3] == 16 && cout == 128))
        return 0;
    FpChainArgs a{};
    a.nt_out = g_fpc_nt;
    a.rows = (int)rows; a.n = n; a.m = m; a.c_skip = c_skip; a.z_stride = z_stride;
    a.z = z_pm; a.skip = skip_pm; a.weight = weight; a.idx = idx; a.wpack = wpack; a.bias = bias;
    a.woff[0] = 0; a.boff[0] = 0; a.woff[1] = dims[0] * dims[1]; a.boff[1] = dims[1];
    a.out = out_pm; a.out_stride = out_stride; a.cout = cout;
    a.hw[0] = hw_a; a.hb[0] = hb_a; a.hw[1] = hw_b; a.hb[1] = hb_b;
    int wo = 0, bo = 0;
    for (int l = 0; l < 3; ++l) {
        a.hwoff[l] = wo; a.hboff[l] = bo;
        wo += hdims[l] * hdims[l + 1];
        bo += hdims[l + 1];
    }
    a.hout[0] = out_a; a.hout_stride[0] = out_stride_a; a.hcout[0] = cout_a;
    a.hout[1] = out_b; a.hout_stride[1] = out_stride_b; a.hcout[1] = cout_b;
    a.hrelu_last = relu_last;
    // tiles per workgroup (consecutive ones, see the kernel): a tile of this kernel is seven layers long, so with the chain kernels'
    // grid cap (3072 workgroups of 3 tiles at the bench shape: 5.3 rounds of 512 resident ones) a sixth of the launch ran with a
    // third of the chip (2.07 ms); short workgroups even out like the pair kernel's do
    const long long tiles = (rows + 63) / 64;
    const long long per8 = (tiles + 7) / 8;
    long long wgs = 8 * ((per8 + g_fph_tiles - 1) / g_fph_tiles);
    if (wgs > (1 << 20)) wgs = 1 << 20;
    const int grid = (int)wgs;
    hipLaunchKernelGGL((fp_chain_kernel<0, 8, 8, true>), dim3(grid), dim3(RC_THREADS), 0, as_stream(stream), a);
    *launched = 1;
    return check_launch("fp_head_fused");
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
    const int grid = (int)(tiles < 256 * g_rc_wg_per_cu ? tiles : 256 * g_rc_wg_per_cu);
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

// Two three-layer chains of equal widths over the same rows in one launch.  Returns 1 in *launched when an instantiation fits.
int rows_chain_pair_launch(void *stream, int rows, int cin, const float *in_pm, int nlayers, const int *dims, const float *wpack_a,
                           const float *bias_a, const float *wpack_b, const float *bias_b, int relu_last, float *out_a, int out_stride_a,
                           int cout_a, float *out_b, int out_stride_b, int cout_b, int *launched) {
    *launched = 0;
    if (rows < 8192 || cin % 16 != 0 || nlayers != 3 || dims[0] != cin) return 0;
    RowsChainPairArgs a{};
    a.rows = rows; a.in_stride = cin; a.in = in_pm;
    a.wpack[0] = wpack_a; a.bias[0] = bias_a; a.wpack[1] = wpack_b; a.bias[1] = bias_b;
    int wo = 0, bo = 0;
    for (int l = 0; l < nlayers; ++l) {
        a.woff[l] = wo; a.boff[l] = bo;
        wo += dims[l] * dims[l + 1];
        bo += dims[l + 1];
    }
    a.out[0] = out_a; a.out_stride[0] = out_stride_a; a.cout[0] = cout_a;
    a.out[1] = out_b; a.out_stride[1] = out_stride_b; a.cout[1] = cout_b;
    a.relu_last = relu_last;
    const long long tiles = ((long long)rows + 63) / 64;
    const int grid = (int)(tiles < 256 * g_rc_wg_per_cu ? tiles : 256 * g_rc_wg_per_cu);
    if (rc_shape_is(nlayers, dims, 8, 16, 16, 1)) {   // point head: 128 -> 256 -> 256 -> <= 16, class logits and box code
        hipLaunchKernelGGL((rows_chain_pair_kernel<8, 16, 16, 1>), dim3(grid), dim3(RC_THREADS), 0, as_stream(stream), a);
        *launched = 1;
        return check_launch("rows_mlp_fused_pair(chain)");
    }
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
    a.dw_H = H; a.dw_W = W; a.dw_by_xcd = g_rc_dw_xcd; a.dw_w = dw_w; a.dw_shift = dw_shift;
    const long long tiles = (long long)B * ((H + 3) / 4) * ((W + 15) / 16);
    const int grid = (int)(tiles < 256 * g_rc_dw_wg_per_cu ? tiles : 256 * g_rc_dw_wg_per_cu);
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

// The backbone's last FP module (first layer's known part pre-applied: z, as pdm_fp_mlp_fused_pre) and the point head's two
// stacks (as pdm_rows_mlp_fused_pair) in ONE launch: out_pm = the module's output rows (point_features), out_a / out_b = the
// stacks' outputs on them.  Bit-identical to pdm_fp_mlp_fused_pre through the chain kernel followed by pdm_rows_mlp_fused_pair.
// Only the shapes rows_chain.hip instantiates (<= 4 skip channels, 128 -> 128; stacks 128 -> 256 -> 256 -> <= 16, >= 32768 rows):
// PDM_E_BADARG otherwise, and the caller issues the two calls.
extern "C" int pdm_fp_head_fused(void *stream, int b, int n, int m, int c_skip, const float *z_pm, int z_stride, const float *skip_pm,
                                 const int *idx, const float *weight, const int *dims, const float *wpack, const float *bias,
                                 float *out_pm, int out_stride, int cout, const int *hdims, const float *hw_a, const float *hb_a,
                                 const float *hw_b, const float *hb_b, int relu_last, float *out_a, int out_stride_a, int cout_a,
                                 float *out_b, int out_stride_b, int cout_b) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 1 && c_skip >= 0, PDM_E_BADARG, "fp_head_fused: bad size");
    if (b == 0 || n == 0) return 0;
    PDM_REQUIRE(z_pm && idx && weight && dims && wpack && bias && out_pm && hdims && hw_a && hb_a && hw_b && hb_b && out_a && out_b &&
                    (c_skip == 0 || skip_pm), PDM_E_BADARG, "fp_head_fused: null pointer");
    PDM_REQUIRE(out_a != out_b && cout_a > 0 && cout_b > 0 && cout_a <= 16 && cout_b <= 16 && cout_a <= out_stride_a && cout_b <= out_stride_b &&
                    out_stride_a % 4 == 0 && out_stride_b % 4 == 0 && out_stride % 4 == 0 && cout <= out_stride && z_stride % 4 == 0 && z_stride >= 128,
                PDM_E_BADARG, "fp_head_fused: strides / widths");
    PDM_REQUIRE(((reinterpret_cast<uintptr_t>(z_pm) | reinterpret_cast<uintptr_t>(wpack) | reinterpret_cast<uintptr_t>(bias) |
                  reinterpret_cast<uintptr_t>(out_pm) | reinterpret_cast<uintptr_t>(hw_a) | reinterpret_cast<uintptr_t>(hb_a) |
                  reinterpret_cast<uintptr_t>(hw_b) | reinterpret_cast<uintptr_t>(hb_b) | reinterpret_cast<uintptr_t>(out_a) |
                  reinterpret_cast<uintptr_t>(out_b)) & 15) == 0, PDM_E_BADARG, "fp_head_fused: buffers must be 16-byte aligned");
    int launched = 0;
    const int rc = pdm::fp_head_launch(stream, b, n, m, c_skip, z_pm, z_stride, skip_pm, idx, weight, dims, wpack, bias, out_pm, out_stride,
                                       cout, hdims, hw_a, hb_a, hw_b, hb_b, relu_last, out_a, out_stride_a, cout_a, out_b, out_stride_b,
                                       cout_b, &launched);
    if (rc) return rc;
    PDM_REQUIRE(launched, PDM_E_BADARG, "fp_head_fused: no instantiation for these shapes");
    return 0;
}
