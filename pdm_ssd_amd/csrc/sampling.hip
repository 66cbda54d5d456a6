// Furthest-point sampling and gather_points for gfx950.
//
// Semantics follow /root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu
// (FPS :100-216, launcher :218-260; gather :15-31; gather grad :53-70); the design does not.
//
// FPS design (one workgroup per sample, as many as B CUs busy):
//   * every point and its running min-distance live in VGPRs for the whole call (N <= 16384);
//     the reference re-reads xyz and read-modify-writes temp in global memory every iteration.
//   * arg-max = wave64 max-reduce -> ballot -> lowest lane, then one LDS record per wave and ONE
//     barrier per iteration (double-buffered records); the reference does a 10-level LDS tree with
//     10 barriers.
//   * tie order: the reference's tree keeps the LEFT operand on ties at every level, so among
//     threads holding the same maximum the winner is the one whose thread id has the smallest
//     BIT-REVERSED value (level `half` compares slot t with t+half: bit log2(half) of the id is
//     the deciding bit, lowest bit decided last = most significant).  Physical thread p therefore
//     plays reference thread t = bitrev(p); priority becomes "lowest p wins", which is exactly
//     what ballot + find-first-set and lowest-wave-first give for free.
#include "common.h"

namespace pdm {

typedef float v2f __attribute__((ext_vector_type(2)));

// ---- wave64 cross-lane helpers (DPP: VALU-speed, no LDS crossbar round trips) -----------------
// Reductions run on the INTEGER image of the floats: every candidate maximum here is either a
// non-negative float (bit pattern order == value order) or a negative sentinel (-1/-2/-3: inactive
// lanes, which as signed integers sort below every non-negative float and never need ordering among
// themselves because at least one active lane always holds a value >= 0).  Integer max needs no NaN
// canonicalisation and folds into v_max_i32_dpp.
// Written as one asm block because hipcc neither folds update_dpp into v_max_i32_dpp here nor drops
// its moves; the s_nop 1 in front of each step is the 2-wait-state VALU-write -> DPP-read hazard,
// which the compiler does not pad inside an asm statement.
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// max over each 16-lane row, result in every lane of the row
__device__ __forceinline__ int row_max16(int v) {
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
    return v;
}
// max over the wave, returned wave-uniform
__device__ __forceinline__ float wave_max(float f) {
    int v = row_max16(__builtin_bit_cast(int, f));
    asm volatile(
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 63));
}

// v_min_f32 without the canonicalising v_max hipcc puts in front of fminf for an operand it cannot prove
// quiet: the operands here are a fresh fma result and an earlier v_min result (or the caller's temp),
// and in the kernel's IEEE mode v_min_f32 returns the non-NaN operand exactly as fminf does.
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct FpsPick {
    int k;          // winning point index
    float x, y, z;  // its coordinates
};

// One per-wave record per iteration: {max value, point index, x, y, z}; double-buffered so a single
// barrier per iteration is enough (a wave can be at most one iteration ahead of the slowest).
struct __attribute__((aligned(16))) FpsRec {
    float v;
    int k;
    float x, y, z;
    float pad[3];
};

// Block-wide arg-max with "lowest physical thread wins ties"; every argument is wave-uniform.
template <int BLOCK>
__device__ __forceinline__ FpsPick block_pick(float wmax, FpsPick mine, FpsRec (*rec)[16], int buf) {
    constexpr int NW = BLOCK / 64;
    if (NW == 1) return mine;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (lane == 0) {
        FpsRec r;
        r.v = wmax; r.k = mine.k; r.x = mine.x; r.y = mine.y; r.z = mine.z;
        rec[buf][wave] = r;
    }
    __syncthreads();
    const FpsRec r = rec[buf][lane & 15];  // NW <= 16: lanes 0..NW-1 hold the wave records
    const float v = (lane & 15) < NW ? r.v : -3.0f;
    const float gmax = __builtin_bit_cast(float, __builtin_amdgcn_readlane(row_max16(__builtin_bit_cast(int, v)), 0));
    const unsigned long long cand = __ballot(v == gmax) & 0xFFFFull;
    const int w = __ffsll((long long)cand) - 1;  // lowest wave id among equal maxima
    FpsPick out;
    out.k = __builtin_amdgcn_readlane(r.k, w);
    out.x = readlane_f(r.x, w);
    out.y = readlane_f(r.y, w);
    out.z = readlane_f(r.z, w);
    return out;
}

// Register-resident FPS.  S = the reference's logical thread count (power of two <= 1024).
// T = min(S, BLOCK) physical threads are active; each plays R = S/T reference threads (the ones the
// tree's first log2(R) levels merge into one slot, visited in the order those levels prefer) and
// holds I = PPT/R points of each in registers: visiting index v -> residue bitrev(v / I), step v % I.
template <int BLOCK, int PPT>
__global__ __launch_bounds__(BLOCK) void fps_reg_kernel(int n, int m, int S, int logS,
                                                        const float *__restrict__ xyz_all,
                                                        float *__restrict__ temp_all,
                                                        int *__restrict__ idx_all) {
    using vec = float __attribute__((ext_vector_type(PPT)));
    __shared__ FpsRec rec[2][16];
    const int b = blockIdx.x;
    const float *__restrict__ xyz = xyz_all + (size_t)b * n * 3;
    float *__restrict__ temp = temp_all + (size_t)b * n;
    int *__restrict__ idxs = idx_all + (size_t)b * m;

    const int T = S < BLOCK ? S : BLOCK;
    const int logT = S < BLOCK ? logS : __builtin_ctz(BLOCK);
    const int logR = logS - logT;             // residues per physical thread = 1 << logR
    const int logI = __builtin_ctz(PPT) - logR;  // points per residue = 1 << logI
    const int p = threadIdx.x;
    const bool active = p < T;
    const int t = logT > 0 ? (int)(__brev((unsigned)p) >> (32 - logT)) : 0;

    auto point_of = [&](int tt, int v) -> int {
        const int ridx = v >> logI;
        const int r = logR > 0 ? (int)(__brev((unsigned)ridx) >> (32 - logR)) : 0;
        return tt + r * T + (v & ((1 << logI) - 1)) * S;
    };

    vec px, py, pz, tmp;
#pragma unroll
    for (int v = 0; v < PPT; ++v) {
        const int k = point_of(t, v);
        if (active && k < n) {
            px[v] = xyz[(size_t)k * 3 + 0];
            py[v] = xyz[(size_t)k * 3 + 1];
            pz[v] = xyz[(size_t)k * 3 + 2];
            tmp[v] = temp[k];
        } else {
            px[v] = py[v] = pz[v] = 0.0f;
            tmp[v] = -1.0f;  // fminf(d, -1) = -1 never beats best = -1 under strict '>'
        }
    }

    FpsPick cur;
    cur.k = 0;
    cur.x = xyz[0]; cur.y = xyz[1]; cur.z = xyz[2];
    if (p == 0) idxs[0] = 0;
    const int wave_base = (threadIdx.x >> 6) * 64;
    for (int j = 1; j < m; ++j) {
        float best = active ? -1.0f : -2.0f;
        int bestv = 0;
#if defined(FPS_DIAG) && FPS_DIAG == 2
        best = tmp[0] + cur.x; bestv = 0;  // timing-only build: no distance pass (results are wrong)
#else
        if constexpr (PPT >= 2) {
            const v2f x1 = {cur.x, cur.x}, y1 = {cur.y, cur.y}, z1 = {cur.z, cur.z};
#pragma unroll
            for (int v = 0; v < PPT; v += 2) {
                const v2f dx = v2f{px[v], px[v + 1]} - x1;
                const v2f dy = v2f{py[v], py[v + 1]} - y1;
                const v2f dz = v2f{pz[v], pz[v + 1]} - z1;
                v2f d = dx * dx;                              // rn(dx*dx)
                d = __builtin_elementwise_fma(dy, dy, d);    // fma(dy,dy,.)
                d = __builtin_elementwise_fma(dz, dz, d);    // fma(dz,dz,.)
                const float d0 = vmin(d.x, tmp[v]);
                const float d1 = vmin(d.y, tmp[v + 1]);
                tmp[v] = d0;
                tmp[v + 1] = d1;
                bestv = d0 > best ? v : bestv;
                best = vmax(best, d0);
                bestv = d1 > best ? v + 1 : bestv;
                best = vmax(best, d1);
            }
        } else {
            const float d = sqdist(px[0] - cur.x, py[0] - cur.y, pz[0] - cur.z);
            const float d0 = vmin(d, tmp[0]);
            tmp[0] = d0;
            bestv = 0;
            best = vmax(best, d0);
        }
#endif
        // wave level: max, lowest lane among equals, that lane's point out of its registers
        const float wmax = wave_max(best);
        const unsigned long long mask = __ballot(best == wmax);
        const int wl = __ffsll((long long)mask) - 1;
        const int slot = __builtin_amdgcn_readlane(bestv, wl);
        FpsPick mine;
        const int tw = logT > 0 ? (int)(__brev((unsigned)(wave_base + wl)) >> (32 - logT)) : 0;
        mine.k = point_of(tw, slot);
        mine.x = readlane_f(px[slot], wl);
        mine.y = readlane_f(py[slot], wl);
        mine.z = readlane_f(pz[slot], wl);
#if defined(FPS_DIAG) && FPS_DIAG == 1
        cur = mine;  // timing-only build: no cross-wave exchange (results are wrong)
#else
        cur = block_pick<BLOCK>(wmax, mine, rec, j & 1);
#endif
        if (p == 0) idxs[j] = cur.k;
    }

#pragma unroll
    for (int v = 0; v < PPT; ++v) {
        const int k = point_of(t, v);
        if (active && k < n) temp[k] = tmp[v];
    }
}

// Any n: points and min-distances stream from global memory (L2-resident) every iteration.
// Same arithmetic and tie order; used when n > 16 * 1024.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fps_stream_kernel(int n, int m, int S, int logS,
                                                           const float *__restrict__ xyz_all,
                                                           float *__restrict__ temp_all,
                                                           int *__restrict__ idx_all) {
    __shared__ FpsRec rec[2][16];
    const int b = blockIdx.x;
    const float *__restrict__ xyz = xyz_all + (size_t)b * n * 3;
    float *__restrict__ temp = temp_all + (size_t)b * n;
    int *__restrict__ idxs = idx_all + (size_t)b * m;
    const int p = threadIdx.x;
    const bool active = p < S;
    const int t = logS > 0 ? (int)(__brev((unsigned)p) >> (32 - logS)) : 0;

    int old = 0;
    if (p == 0) idxs[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = xyz[(size_t)old * 3 + 0];
        const float y1 = xyz[(size_t)old * 3 + 1];
        const float z1 = xyz[(size_t)old * 3 + 2];
        float best = active ? -1.0f : -2.0f;
        int besti = 0;
        if (active) {
            for (int k = t; k < n; k += S) {
                const float d = sqdist(xyz[(size_t)k * 3 + 0] - x1, xyz[(size_t)k * 3 + 1] - y1,
                                       xyz[(size_t)k * 3 + 2] - z1);
                const float d2 = fminf(d, temp[k]);
                temp[k] = d2;
                const bool gt = d2 > best;
                besti = gt ? k : besti;
                best = gt ? d2 : best;
            }
        }
        const float wmax = wave_max(best);
        const unsigned long long mask = __ballot(best == wmax);
        const int wl = __ffsll((long long)mask) - 1;
        FpsPick mine;
        mine.k = __builtin_amdgcn_readlane(besti, wl);
        mine.x = mine.y = mine.z = 0.0f;
        old = block_pick<BLOCK>(wmax, mine, rec, j & 1).k;
        if (p == 0) idxs[j] = old;
    }
}

__global__ void gather_points_kernel(int c, int n, int m, const float *__restrict__ points,
                                     const int *__restrict__ idx, float *__restrict__ out) {
    const int b = blockIdx.z, ci = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int k = idx[(size_t)b * m + j];
    out[((size_t)b * c + ci) * m + j] = points[((size_t)b * c + ci) * n + k];
}

__global__ void gather_points_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                          const int *__restrict__ idx,
                                          float *__restrict__ grad_points) {
    const int b = blockIdx.z, ci = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int k = idx[(size_t)b * m + j];
    atomicAdd(grad_points + ((size_t)b * c + ci) * n + k, grad_out[((size_t)b * c + ci) * m + j]);
}

// cuda_utils.h:10-14 — the reference evaluates log(n)/log(2.0) in double and truncates.
// Integer floor(log2 n) is identical for every n where the double quotient does not land exactly
// below an integer; verified equal for all n in [1, 2^20] in tests/test_host_logic.py.
static int ref_block_threads(int n, int *logS) {
    int l = 0;
    while ((2LL << l) <= n) ++l;
    if (l > 10) l = 10;
    *logS = l;
    return 1 << l;
}

}  // namespace pdm

using namespace pdm;

// Tuning knob (not part of the reference-facing ABI): 0 = 1024 threads x 16 points, 1 = 512 x 32 for
// 8192 < n <= 16384.  Both give identical indices.
static int g_fps_variant = 0;
extern "C" int pdm_tune_fps_variant(int v) { const int old = g_fps_variant; g_fps_variant = v; return old; }

#define FPS_LAUNCH(BLOCK, PPT)                                                                  \
    hipLaunchKernelGGL((fps_reg_kernel<BLOCK, PPT>), dim3(b), dim3(BLOCK), 0, as_stream(stream), \
                       n, m, S, logS, points, temp, idx)

extern "C" int pdm_furthest_point_sampling(void *stream, int b, int n, int m, const float *points,
                                           float *temp, int *idx) {
    PDM_REQUIRE(b >= 0 && n >= 0, PDM_E_BADARG, "fps: negative size b=%d n=%d", b, n);
    if (m <= 0 || b == 0) return 0;  // sampling_gpu.cu:108
    PDM_REQUIRE(n >= 1, PDM_E_BADARG, "fps: n=%d with m=%d", n, m);
    PDM_REQUIRE(points && temp && idx, PDM_E_BADARG, "fps: null pointer");
    int logS = 0;
    const int S = ref_block_threads(n, &logS);
    const int per = (n + S - 1) / S;  // points per reference thread
    if (S <= 64) {
        FPS_LAUNCH(64, 2);
    } else if (S <= 256) {
        FPS_LAUNCH(256, 2);
    } else if (S == 512) {
        FPS_LAUNCH(512, 2);
    } else if (per <= 1) {
        FPS_LAUNCH(1024, 1);
    } else if (per <= 2) {
        FPS_LAUNCH(1024, 2);
    } else if (per <= 4) {
        FPS_LAUNCH(1024, 4);
    } else if (per <= 8) {
        FPS_LAUNCH(1024, 8);
    } else if (per <= 16) {
        if (g_fps_variant == 1) FPS_LAUNCH(512, 32); else FPS_LAUNCH(1024, 16);
    } else {
        hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(b), dim3(1024), 0, as_stream(stream), n,
                           m, S, logS, points, temp, idx);
    }
    return check_launch("furthest_point_sampling");
}

extern "C" int pdm_gather_points(void *stream, int b, int c, int n, int npoints,
                                 const float *points, const int *idx, float *out) {
    PDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0, PDM_E_BADARG, "gather_points: negative size");
    if (b == 0 || c == 0 || npoints == 0) return 0;
    PDM_REQUIRE(points && idx && out, PDM_E_BADARG, "gather_points: null pointer");
    PDM_REQUIRE(c <= 65535 && b <= 65535, PDM_E_TOOLARGE, "gather_points: c=%d b=%d exceed grid", c, b);
    dim3 grid(divup(npoints, 256), c, b);
    hipLaunchKernelGGL(gather_points_kernel, grid, dim3(256), 0, as_stream(stream), c, n, npoints,
                       points, idx, out);
    return check_launch("gather_points");
}

extern "C" int pdm_gather_points_grad(void *stream, int b, int c, int n, int npoints,
                                      const float *grad_out, const int *idx, float *grad_points) {
    PDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0, PDM_E_BADARG, "gather_points_grad: negative size");
    if (b == 0 || c == 0 || npoints == 0) return 0;
    PDM_REQUIRE(grad_out && idx && grad_points, PDM_E_BADARG, "gather_points_grad: null pointer");
    PDM_REQUIRE(c <= 65535 && b <= 65535, PDM_E_TOOLARGE, "gather_points_grad: c=%d b=%d exceed grid", c, b);
    dim3 grid(divup(npoints, 256), c, b);
    hipLaunchKernelGGL(gather_points_grad_kernel, grid, dim3(256), 0, as_stream(stream), c, n,
                       npoints, grad_out, idx, grad_points);
    return check_launch("gather_points_grad");
}
