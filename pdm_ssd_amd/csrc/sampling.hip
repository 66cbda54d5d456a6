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

// Ragged ("stacked") batches (pointnet2_stack/src/sampling_gpu.cu:187-319): per-sample point and sample counts,
// samples packed back to back, GLOBAL indices out; the reference always runs 1024 logical threads there.
struct FpsRagged {
    const int *cnt;    // null = batch mode
    const int *mcnt;
};
struct FpsWhere { int start, ostart, n, m; };
__device__ __forceinline__ FpsWhere fps_where(const FpsRagged &rg, int b, int n, int m) {
    FpsWhere w;
    if (rg.cnt) {
        int s = 0, o = 0;
        for (int k = 0; k < b; ++k) { s += rg.cnt[k]; o += rg.mcnt[k]; }
        w.start = s; w.ostart = o; w.n = rg.cnt[b]; w.m = rg.mcnt[b];
    } else {
        w.start = b * n; w.ostart = b * m; w.n = n; w.m = m;
    }
    return w;
}

// Register-resident FPS.  S = the reference's logical thread count (power of two <= 1024).
// T = min(S, BLOCK) physical threads are active; each plays R = S/T reference threads (the ones the
// tree's first log2(R) levels merge into one slot, visited in the order those levels prefer) and
// holds I = PPT/R points of each in registers: visiting index v -> residue bitrev(v / I), step v % I.
template <int BLOCK, int PPT>
__global__ __launch_bounds__(BLOCK) void fps_reg_kernel(int n_, int m_, int S, int logS,
                                                        const float *__restrict__ xyz_all,
                                                        float *__restrict__ temp_all,
                                                        int *__restrict__ idx_all, FpsRagged rg) {
    using vec = float __attribute__((ext_vector_type(PPT)));
    __shared__ FpsRec rec[2][16];
    const int b = blockIdx.x;
    const FpsWhere wh = fps_where(rg, b, n_, m_);
    const int n = wh.n, m = wh.m;
    const int gofs = rg.cnt ? wh.start : 0;   // stacked batches report global indices
    if (m <= 0) return;                       // whole workgroup
    const float *__restrict__ xyz = xyz_all + (size_t)wh.start * 3;
    float *__restrict__ temp = temp_all + (size_t)wh.start;
    int *__restrict__ idxs = idx_all + (size_t)wh.ostart;

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
    if (p == 0) idxs[0] = gofs;
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
        if (p == 0) idxs[j] = cur.k + gofs;
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
__global__ __launch_bounds__(BLOCK) void fps_stream_kernel(int n_, int m_, int S, int logS,
                                                           const float *__restrict__ xyz_all,
                                                           float *__restrict__ temp_all,
                                                           int *__restrict__ idx_all, FpsRagged rg) {
    __shared__ FpsRec rec[2][16];
    const int b = blockIdx.x;
    const FpsWhere wh = fps_where(rg, b, n_, m_);
    const int n = wh.n, m = wh.m;
    const int gofs = rg.cnt ? wh.start : 0;
    if (m <= 0) return;
    const float *__restrict__ xyz = xyz_all + (size_t)wh.start * 3;
    float *__restrict__ temp = temp_all + (size_t)wh.start;
    int *__restrict__ idxs = idx_all + (size_t)wh.ostart;
    const int p = threadIdx.x;
    const bool active = p < S;
    const int t = logS > 0 ? (int)(__brev((unsigned)p) >> (32 - logS)) : 0;

    int old = 0;
    if (p == 0) idxs[0] = gofs;
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
        if (p == 0) idxs[j] = old + gofs;
    }
}

// ------------------------------------------------------------------------------------------------
// Pruned FPS (exact).  Same selection as fps_reg_kernel / the reference, for 1024 < n <= 1024 * PPT.
//
// Observation: adding sample c lowers temp[p] only where d(p, c) < temp[p].  If the points a wave
// holds lie in a box B and dist^2(c, B) >= the wave's largest temp, none of them changes and the
// wave's (max, arg-max) of the previous iteration is still valid: the wave skips its distance pass.
// To make that happen often, the workgroup first sorts its cloud by a 12-bit Morton cell code (LDS
// counting sort) and deals contiguous runs of 64*PPT sorted points to the waves, so each wave owns a
// compact region.  After the first few hundred samples only the one to three waves near c compute.
//
// Tie order without positional tricks: a point's priority among equal maxima is
//     rank(k) = bitrev10(k mod 1024) * 16 + k div 1024       (the reference's tree order, see above);
// each lane sorts its PPT points by rank once, so strict '>' in slot order is right inside a lane;
// between lanes and between waves ties are rare and resolved explicitly by comparing ranks.
// The skip test is conservative under rounding: the bound is scaled by (1 - 2^-18), far more than the
// <= 5 roundings (2^-24 each) that separate it from any computed distance.
__device__ __forceinline__ int row_min16_i(int v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = row_min16_i(v);
    asm volatile(
        "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int fps_rank1024(int k) {  // reference priority for block size 1024
    return (int)((__brev((unsigned)(k & 1023)) >> 22) << 4) | (k >> 10);
}

__device__ __forceinline__ unsigned morton12(unsigned cx, unsigned cy, unsigned cz) {
    unsigned key = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        key |= ((cx >> b) & 1u) << (3 * b) | ((cy >> b) & 1u) << (3 * b + 1) | ((cz >> b) & 1u) << (3 * b + 2);
    return key;
}

struct __attribute__((aligned(16))) FpsRecR {
    float v;
    int k;
    float x, y, z;
    int rank;
    float pad[2];
};

// MULTI: a cloud of more than BLOCK*PPT points is split over G workgroups (contiguous index ranges, each
// register-resident as above); every iteration each workgroup publishes its local winner as six 8-byte
// {payload, iteration} granules (agent-scope relaxed atomics = sc1 stores/loads, the data-tagged hand-off of
// the CDNA guide: no flag, no fence) and picks the best of the G records.  All G workgroups of a cloud must be
// resident together: the host launches at most 256 workgroups at a time.  Every spin is bounded.
// Resumable segments (pdm_furthest_point_sampling_jobs): up to 4 jobs in one launch, job q = iterations [j0, j1) of
// the FPS of its own batch of `nb` clouds, continuing from the state an earlier segment left in temp (running
// min-distances) and idx (samples [0, j0)).  njobs == 0: the kernel's plain arguments, iterations [1, m).
struct FpsSeg {
    int njobs, nb;
    const float *xyz[4];
    float *temp[4];
    int *idx[4];
    unsigned long long *xch[4];   // MULTI: each job's own exchange slots
    unsigned *status[4];          // MULTI: raised when a workgroup gave up waiting for a peer ([0] also for njobs == 0)
    int j0[4], j1[4];
};
#define FPS_SEG_PICK(F, q) ((q) == 0 ? seg.F[0] : (q) == 1 ? seg.F[1] : (q) == 2 ? seg.F[2] : seg.F[3])

#if defined(FPS_DIAG) && FPS_DIAG == 4
// Phase-timing build (tools/diag/fps_phase.py): s_memtime stamps of workgroup 0 around the phases of an iteration,
// summed separately for waves that ran a distance pass in that iteration (row 1) and waves that skipped (row 0).
// [row][0..6] = bound+ballot, passes, record write, barrier wait, record read (LDS), decode, whole iteration;
// [row][7] = count.  Results stay exact; the stamps cost ~10 % (each waits for outstanding LDS/scalar loads).
__device__ unsigned long long g_fps_phase[2][8];
#define FPS_STAMP(t) const unsigned long long t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f)
#else
#define FPS_STAMP(t)
#endif

// SUB > 1: a lane's PPT points are SUB separate groups of GP = PPT / SUB points; group s of wave w is run
// (s * NW + w) * 64 * GP .. of the sorted order, so a wave owns SUB compact boxes that are NOT neighbours, each with
// its own cached (max, arg-max) record; the skip test runs per group (lane s tests group s, all in one pass of
// instructions) and a sample near one box costs a pass over GP points per lane instead of PPT.  What an iteration
// costs is, per SIMD, (waves on it) x (record decode + skip test + record write, ~55 instructions whatever the
// wave holds) + the passes that do run; fewer, fatter waves with finer groups cut both terms
// (profiles/r02_fps_phase_table.md).
template <int BLOCK, int PPT, int SUB, bool MULTI>
__global__ __launch_bounds__(BLOCK) void fps_pruned_kernel(int n_total, int m, int G,
                                                          const float *__restrict__ xyz_arg,
                                                          float *__restrict__ temp_arg,
                                                          int *__restrict__ idx_arg,
                                                          unsigned long long *__restrict__ xch_arg,
                                                          const FpsSeg seg) {
    const bool segd = seg.njobs > 0;
    const int per_job = segd ? seg.nb * (MULTI ? G : 1) : (int)gridDim.x;   // workgroups of one job
    const int job = segd ? (int)blockIdx.x / per_job : 0;   // static indices only: a runtime index would spill the struct
    const int blk = (int)blockIdx.x - job * per_job;        // workgroup number inside the job
    unsigned long long *__restrict__ xch_all = segd ? FPS_SEG_PICK(xch, job) : xch_arg;
    unsigned *status_word = segd ? FPS_SEG_PICK(status, job) : seg.status[0];
    const float *__restrict__ xyz_all = segd ? FPS_SEG_PICK(xyz, job) : xyz_arg;
    float *__restrict__ temp_all = segd ? FPS_SEG_PICK(temp, job) : temp_arg;
    int *__restrict__ idx_all = segd ? FPS_SEG_PICK(idx, job) : idx_arg;
    const int jb0 = segd ? FPS_SEG_PICK(j0, job) : 1;
    const int jb1 = segd ? FPS_SEG_PICK(j1, job) : m;
    constexpr int NW = BLOCK / 64, HPT = 4096 / BLOCK;  // waves, histogram bins per thread
    constexpr int GP = PPT / SUB, NG = NW * SUB;         // points per lane and group, groups per workgroup
    static_assert(GP >= 2 && GP * SUB == PPT && (GP & (GP - 1)) == 0 && NG <= 64, "fps_pruned_kernel: group shape");
    using vec = float __attribute__((ext_vector_type(PPT)));
    __shared__ float4 recA[2][NG];   // {max, x, y, z} of a group's arg-max
    __shared__ int2 recB[2][NG];     // {iteration it was written in, handle}
    __shared__ int hist[4096];
    __shared__ unsigned short order[BLOCK * PPT];  // sorted point indices, then (slot, thread) -> index
    __shared__ float red[6][NW];
    __shared__ int wsum[NW];
    // MULTI: workgroups are dealt to the 8 XCDs round-robin by blockIdx; the G workgroups of one cloud take ids
    // b, b + nb, b + 2 nb, ... so that (when nb is a multiple of 8) they share an XCD and its L2 for the exchange
    const int nb_ = MULTI ? per_job / G : 1;
    const int b = MULTI ? blk % nb_ : blk;
    const int grp = MULTI ? blk / nb_ : 0;
    const int per = MULTI ? (n_total + G - 1) / G : n_total;   // points per workgroup (<= BLOCK*PPT)
    const int k0 = grp * per;                                   // first global index of this workgroup
    const int n = max(0, min(per, n_total - k0));               // its point count
    const int rankQ = (n_total + 1023) >> 10;                   // points per reference thread (block size 1024)
    const float *__restrict__ xyz0 = xyz_all + (size_t)b * n_total * 3;  // the cloud
    const float *__restrict__ xyz = xyz0 + (size_t)k0 * 3;               // this workgroup's slice
    float *__restrict__ temp = temp_all + (size_t)b * n_total + k0;
    int *__restrict__ idxs = idx_all + (size_t)b * m;
    const int p = threadIdx.x, lane = p & 63, wave = __builtin_amdgcn_readfirstlane(p >> 6);
    auto rank_of = [&](int klocal) -> int {  // reference priority of global index k0 + klocal
        const int kg = k0 + klocal;
        return (int)(__brev((unsigned)(kg & 1023)) >> 22) * rankQ + (kg >> 10);
    };

    // ---- bounding box of the cloud
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = p; k < n; k += BLOCK)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[(size_t)k * 3 + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off >= 1; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
        }
        if (lane == 0) { red[a][wave] = mn[a]; red[3 + a][wave] = mx[a]; }
    }
    for (int c = p; c < 4096; c += BLOCK) hist[c] = 0;
    __syncthreads();
    // 12 key bits, dealt one at a time to the axis whose cells are currently longest, most significant first:
    // the top bits split the long axes, so every run of n/16 sorted points is a compact box even for flat
    // scenes (a plain 4+4+4 Morton code would slice a 70 x 80 x 4 m cloud into thin z layers).
    float lo[3], sc[3];
    int nbits[3] = {0, 0, 0};
    unsigned seq = 0;  // 2 bits per step: axis of key bit 11, 10, ...
    {
        float ext[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float l = red[a][0], h = red[3 + a][0];
            for (int w = 1; w < NW; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
            if (!(l <= h)) { l = 0.0f; h = 0.0f; }
            lo[a] = l;
            ext[a] = h - l;
            if (!(ext[a] < INFINITY)) ext[a] = 0.0f;
        }
        float cur_ext[3] = {ext[0], ext[1], ext[2]};
        for (int s_ = 0; s_ < 12; ++s_) {
            int a = cur_ext[0] >= cur_ext[1] ? (cur_ext[0] >= cur_ext[2] ? 0 : 2) : (cur_ext[1] >= cur_ext[2] ? 1 : 2);
            seq |= (unsigned)a << (2 * s_);
            nbits[a] += 1;
            cur_ext[a] *= 0.5f;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) sc[a] = ext[a] > 0.0f ? (float)(1 << nbits[a]) / ext[a] : 0.0f;
    }
    auto key_of = [&](int k) -> unsigned {
        unsigned c[3];
        int used[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float t = (xyz[(size_t)k * 3 + a] - lo[a]) * sc[a];
            c[a] = (unsigned)fminf(fmaxf(t, 0.0f), (float)((1 << nbits[a]) - 1));  // NaN -> 0
        }
        unsigned key = 0;
        for (int s_ = 0; s_ < 12; ++s_) {
            const int a = (seq >> (2 * s_)) & 3;
            used[a] += 1;
            key = (key << 1) | ((c[a] >> (nbits[a] - used[a])) & 1u);
        }
        return key;
    };
    // ---- counting sort by Morton cell (order inside a cell is irrelevant)
    for (int k = p; k < n; k += BLOCK) atomicAdd(&hist[key_of(k)], 1);
    __syncthreads();
    {
        int local[HPT], sum = 0;
#pragma unroll
        for (int i = 0; i < HPT; ++i) { local[i] = hist[p * HPT + i]; sum += local[i]; }
        int incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        if (wave == 0) {
            const int v = lane < NW ? wsum[lane] : 0;
            int inc = v;
            for (int off = 1; off < NW; off <<= 1) {
                const int t = __shfl_up(inc, off, 64);
                if (lane >= off) inc += t;
            }
            if (lane < NW) wsum[lane] = inc - v;
        }
        __syncthreads();
        int run = wsum[wave] + incl - sum;
#pragma unroll
        for (int i = 0; i < HPT; ++i) { hist[p * HPT + i] = run; run += local[i]; }
    }
    __syncthreads();
    for (int k = p; k < n; k += BLOCK) order[atomicAdd(&hist[key_of(k)], 1)] = (unsigned short)k;
    __syncthreads();

    // ---- this lane's PPT points: a contiguous run of the sorted order, then sorted by reference rank
    vec px, py, pz;   // ext vectors: the winner's coordinates are fetched with a wave-uniform dynamic index
    float tmp[PPT];   // plain scalars: only ever indexed statically, and updated on one branch of the loop
    int kk[PPT], rk[PPT];
#pragma unroll
    for (int v = 0; v < PPT; ++v) {
        const int sp = (((v / GP) * NW + wave) * 64 + lane) * GP + (v % GP);
        if (sp < n) {
            const int k = order[sp];
            kk[v] = k;
            rk[v] = rank_of(k);
            px[v] = xyz[(size_t)k * 3 + 0];
            py[v] = xyz[(size_t)k * 3 + 1];
            pz[v] = xyz[(size_t)k * 3 + 2];
            tmp[v] = temp[k];
        } else {
            kk[v] = 0xFFFF;
            rk[v] = 0x7FFFFFFF;  // invalid slots sort last
            px[v] = py[v] = pz[v] = 0.0f;
            tmp[v] = -1.0f;      // fminf(d, -1) = -1 never beats a real distance
        }
    }
#pragma unroll
    for (int k2 = 2; k2 <= GP; k2 <<= 1)   // rank order inside each group of GP slots
#pragma unroll
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1)
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int l = i ^ j2;
                if (l > i) {
                    const bool up = ((i & (GP - 1)) & k2) == 0;
                    const bool sw = (rk[i] > rk[l]) == up;
                    const int tr = rk[i], tk = kk[i];
                    const float tx = px[i], ty = py[i], tz = pz[i], tt = tmp[i];
                    rk[i] = sw ? rk[l] : tr; rk[l] = sw ? tr : rk[l];
                    kk[i] = sw ? kk[l] : tk; kk[l] = sw ? tk : kk[l];
                    px[i] = sw ? px[l] : tx; px[l] = sw ? tx : px[l];
                    py[i] = sw ? py[l] : ty; py[l] = sw ? ty : py[l];
                    pz[i] = sw ? pz[l] : tz; pz[l] = sw ? tz : pz[l];
                    tmp[i] = sw ? tmp[l] : tt; tmp[l] = sw ? tt : tmp[l];
                }
            }
    __syncthreads();  // every thread has read its run of `order`
#pragma unroll
    for (int v = 0; v < PPT; ++v) order[v * BLOCK + p] = (unsigned short)kk[v];

    // ---- boxes of the groups: lane g < NG of the OWNING wave keeps the box of group g = s * NW + wave
    float bl0 = 0.0f, bl1 = 0.0f, bl2 = 0.0f, bh0 = 0.0f, bh1 = 0.0f, bh2 = 0.0f;
#pragma unroll
    for (int s_ = 0; s_ < SUB; ++s_) {
        float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int v = s_ * GP; v < (s_ + 1) * GP; ++v)
            if (kk[v] != 0xFFFF) {
                l[0] = fminf(l[0], px[v]); h[0] = fmaxf(h[0], px[v]);
                l[1] = fminf(l[1], py[v]); h[1] = fmaxf(h[1], py[v]);
                l[2] = fminf(l[2], pz[v]); h[2] = fmaxf(h[2], pz[v]);
            }
#pragma unroll
        for (int a = 0; a < 3; ++a)
            for (int off = 32; off >= 1; off >>= 1) {
                l[a] = fminf(l[a], __shfl_xor(l[a], off, 64));
                h[a] = fmaxf(h[a], __shfl_xor(h[a], off, 64));
            }
        if (lane == s_ * NW + wave) { bl0 = l[0]; bl1 = l[1]; bl2 = l[2]; bh0 = h[0]; bh1 = h[1]; bh2 = h[2]; }
    }
    // group records: written by the lane that holds a group's new arg-max in the iteration the group was passed over,
    // tagged with that iteration; every wave keeps a register copy of all NG records (lane g <-> group g) and takes a
    // record from LDS only when its tag is the current iteration, so unchanged groups cost nothing to republish
    if (p < 2 * NG) reinterpret_cast<int2 *>(recB)[p] = make_int2(-1, 0);
    __syncthreads();

    // ---- iterations
    FpsPick cur;
    cur.k = jb0 > 1 ? idxs[jb0 - 1] : 0;   // a later segment continues from the last sample of the one before
    cur.x = xyz0[(size_t)cur.k * 3]; cur.y = xyz0[(size_t)cur.k * 3 + 1]; cur.z = xyz0[(size_t)cur.k * 3 + 2];
    if (p == 0 && grp == 0 && jb0 <= 1) idxs[0] = 0;
    __shared__ float fin[8];
    const bool own = lane < NG && (lane % NW) == wave;   // this lane stands for one of this wave's groups
    const unsigned long long ownmask = __ballot(own);
    bool first = true;
    float rv = -3.0f, rx = 0.0f, ry = 0.0f, rz = 0.0f;   // register copy of record `lane` (lanes >= NG never win)
    int rh = 0;                                           // its handle: index into order[] (slot * BLOCK + thread)
    int pend_j = -1, pend_k = 0;                          // sample whose index is still to be stored
    for (int j = jb0; j < jb1; ++j) {
        FPS_STAMP(t0);
        // lower bound of every computed distance between cur and a point of the group's box (lane g: group g)
        const float gx = fmaxf(0.0f, fmaxf(bl0 - cur.x, cur.x - bh0));
        const float gy = fmaxf(0.0f, fmaxf(bl1 - cur.y, cur.y - bh1));
        const float gz = fmaxf(0.0f, fmaxf(bl2 - cur.z, cur.z - bh2));
        const float lb = (gx * gx + gy * gy + gz * gz) * 0.99999618530273438f;  // 1 - 2^-18
#if defined(FPS_DIAG) && FPS_DIAG == 2
        const unsigned long long need = first ? ownmask : 0ull;  // timing-only build: passes in the first iteration only
#else
        // a group is passed over unless the bound says nothing in it can change; also when anything is NaN
        const unsigned long long need = first ? ownmask : __ballot(own && !(lb >= rv));
#endif
        first = false;
        const int buf = j & 1;
        FPS_STAMP(t1);
        const unsigned long long mine = need >> wave;   // bit s * NW <-> this wave's group s
        if (mine != 0ull) {
#pragma unroll
        for (int s_ = 0; s_ < SUB; ++s_) {
            if ((mine >> (s_ * NW)) & 1ull) {  // wave-uniform
#if defined(FPS_DIAG) && FPS_DIAG == 3
                if (lane == 0 && b == 0) atomicAdd(&temp_all[(size_t)gridDim.x * n - 1 - (j >> 8)], 1.0f);  // diag only
#endif
                // NC independent (max, first arg-max) chains over contiguous runs of the group's slots, merged with
                // strict '>' in slot order: the chain of one compare + select + max per point is the latency of a pass
                constexpr int NC = GP >= 4 ? 2 : 1, CL = GP / NC;
                float cb[NC];
                int cv[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) { cb[c] = -1.0f; cv[c] = s_ * GP + c * CL; }
                const v2f x1 = {cur.x, cur.x}, y1 = {cur.y, cur.y}, z1 = {cur.z, cur.z};
#pragma unroll
                for (int i = 0; i < CL; i += 2) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int v = s_ * GP + c * CL + i;
                        const v2f dx = v2f{px[v], px[v + 1]} - x1;
                        const v2f dy = v2f{py[v], py[v + 1]} - y1;
                        const v2f dz = v2f{pz[v], pz[v + 1]} - z1;
                        v2f d = dx * dx;
                        d = __builtin_elementwise_fma(dy, dy, d);
                        d = __builtin_elementwise_fma(dz, dz, d);
                        const float d0 = vmin(d.x, tmp[v]);
                        const float d1 = vmin(d.y, tmp[v + 1]);
                        tmp[v] = d0;
                        tmp[v + 1] = d1;
                        cv[c] = d0 > cb[c] ? v : cv[c];
                        cb[c] = vmax(cb[c], d0);
                        cv[c] = d1 > cb[c] ? v + 1 : cv[c];
                        cb[c] = vmax(cb[c], d1);
                    }
                }
                float best = cb[0];
                int bestv = cv[0];
#pragma unroll
                for (int c = 1; c < NC; ++c) {
                    bestv = cb[c] > best ? cv[c] : bestv;
                    best = vmax(best, cb[c]);
                }
                const float wmax = wave_max(best);
                const unsigned long long cand = __ballot(best == wmax);
                int wl = __ffsll((long long)cand) - 1;
                if (__popcll(cand) > 1) {  // equal maxima in several lanes: the reference order decides
                    const int myk = order[bestv * BLOCK + p];
                    const int r = best == wmax ? rank_of(myk) : 0x7FFFFFFF;
                    const int rmin = wave_min_i(r);
                    wl = __ffsll((long long)__ballot(r == rmin)) - 1;
                }
                const int slot = __builtin_amdgcn_readlane(bestv, wl);
                const float fx = px[slot], fy = py[slot], fz = pz[slot];   // wave-uniform register index
                if (lane == wl) {
                    recA[buf][s_ * NW + wave] = make_float4(best, fx, fy, fz);
                    recB[buf][s_ * NW + wave] = make_int2(j, slot * BLOCK + p);
                }
            }
        }
        }
        FPS_STAMP(t2);
        FPS_STAMP(t3);
        __syncthreads();
        FPS_STAMP(t4);
        {
            const float4 na = recA[buf][lane < NG ? lane : 0];
            const int2 nb = recB[buf][lane < NG ? lane : 0];
#if defined(FPS_DIAG) && FPS_DIAG == 4
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            if (lane < NG && nb.x == j) { rv = na.x; rx = na.y; ry = na.z; rz = na.w; rh = nb.y; }
        }
        FPS_STAMP(t5);
        // the sample before this one: its index was requested from LDS an iteration ago
        if (!MULTI && p == 0 && pend_j >= 0) idxs[pend_j] = pend_k;
        float gmax;
        unsigned long long c2;
        int w;
        if constexpr (NG <= 16) {   // (every row of 16 lanes would do; lanes >= NG hold -3)
            gmax = __builtin_bit_cast(float, __builtin_amdgcn_readlane(row_max16(__builtin_bit_cast(int, rv)), 0));
            c2 = __ballot(rv == gmax) & 0xFFFFull;
        } else {
            gmax = wave_max(rv);
            c2 = __ballot(rv == gmax);
        }
        w = __ffsll((long long)c2) - 1;
        if (__popcll(c2) > 1) {   // equal maxima in several groups: the reference order decides
            const int rr = rv == gmax ? rank_of(order[rh]) : 0x7FFFFFFF;
            const int rmin = wave_min_i(rr);
            w = __ffsll((long long)__ballot(rr == rmin)) - 1;
        }
        cur.x = readlane_f(rx, w);
        cur.y = readlane_f(ry, w);
        cur.z = readlane_f(rz, w);
        const int wh = __builtin_amdgcn_readlane(rh, w);
        const int wk = order[wh];           // local index of the sample (LDS, uniform address)
        if (!MULTI) { pend_j = j; pend_k = wk + k0; }
        else cur.k = wk + k0;               // global index
#if defined(FPS_DIAG) && FPS_DIAG == 4
        {
            FPS_STAMP(t6);
            if (blockIdx.x == 0 && lane == 0 && j > jb0 + 8) {
                unsigned long long *row = g_fps_phase[need ? 1 : 0];
                atomicAdd(row + 0, t1 - t0); atomicAdd(row + 1, t2 - t1); atomicAdd(row + 2, t3 - t2);
                atomicAdd(row + 3, t4 - t3); atomicAdd(row + 4, t5 - t4); atomicAdd(row + 5, t6 - t5);
                atomicAdd(row + 6, t6 - t0); atomicAdd(row + 7, 1ull);
            }
        }
#endif
        if (MULTI) {
            const int crank = rank_of(wk);
            unsigned long long *slot = xch_all + ((size_t)b * 2 + (j & 1)) * (size_t)G * 6;
            if (wave == 0) {
                if (lane < 6) {
                    const unsigned payload = lane == 0 ? __float_as_uint(n > 0 ? gmax : -1.0f)
                                           : lane == 1 ? (unsigned)cur.k
                                           : lane == 2 ? __float_as_uint(cur.x)
                                           : lane == 3 ? __float_as_uint(cur.y)
                                           : lane == 4 ? __float_as_uint(cur.z) : (unsigned)crank;
                    __hip_atomic_store(slot + grp * 6 + lane, ((unsigned long long)(unsigned)j << 32) | payload,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                unsigned got = 0;
                if (lane < 6 * G) {
                    bool seen = false;
                    for (int spin = 0; spin < (1 << 16); ++spin) {  // bounded: a lost peer ends the call, not the GPU
                        const unsigned long long v = __hip_atomic_load(slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        got = (unsigned)v;
                        if ((unsigned)(v >> 32) == (unsigned)j) { seen = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    // a peer never published (not co-resident, or the device is shared): the indices from here on
                    // are wrong.  They stay in range (stale records hold earlier winners), and the status word
                    // behind the exchange slots tells the host, which raises (pdm_furthest_point_sampling_status).
                    if (!seen && status_word) atomicOr(status_word, 1u);
                }
                // group q's record sits in lanes 6q .. 6q+5; pick max value, then min rank
                float bv = -2.0f; int bq = 0, br = 0x7FFFFFFF;
                for (int q = 0; q < G; ++q) {
                    const float v = __uint_as_float(__builtin_amdgcn_readlane((int)got, q * 6));
                    const int rr = __builtin_amdgcn_readlane((int)got, q * 6 + 5);
                    if (v > bv || (v == bv && rr < br)) { bv = v; bq = q; br = rr; }
                }
                if (lane == 0) {
                    fin[0] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)got, bq * 6 + 1));
                    fin[1] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)got, bq * 6 + 2));
                    fin[2] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)got, bq * 6 + 3));
                    fin[3] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)got, bq * 6 + 4));
                }
            }
            __syncthreads();
            cur.k = (int)__float_as_uint(fin[0]);
            cur.x = fin[1]; cur.y = fin[2]; cur.z = fin[3];
            __syncthreads();  // fin is rewritten next iteration
        }
        if (MULTI && p == 0 && grp == 0) idxs[j] = cur.k;
    }
    if (!MULTI && p == 0 && pend_j >= 0) idxs[pend_j] = pend_k;

#pragma unroll
    for (int v = 0; v < PPT; ++v) {
        const int k = order[v * BLOCK + p];
        if (k != 0xFFFF) temp[k] = tmp[v];
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

// Tuning knob (not part of the reference-facing ABI).  8192 < n <= 16384: 0 = default (FPS_L1_DEFAULT below);
// 1 = 512 x 32 and 2 = 1024 x 16 without pruning; pruned forms (threads x points per lane / groups per lane):
// 3 = 512x32/1, 4 = 1024x16/1 (round 1's shape), 5 = 1024x16/2, 6 = 512x32/2, 7 = 512x32/4, 9 = 512x32/8, 10 = 1024x16/4.
// 1024 <= n <= 8192: 8 = one physical thread per reference thread without pruning, 16 = round 1's register kernels
// (256 / 512 threads, no pruning), 17 / 18 / 19 = pruned with 1 / 2 / 4 groups per lane.  All give identical indices.
static int g_fps_variant = 0;
#if defined(FPS_DIAG) && FPS_DIAG == 4
extern "C" int pdm_fps_phase_read(unsigned long long *out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(pdm::g_fps_phase), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(pdm::g_fps_phase), z, sizeof(z));
    }
    return (int)e;
}
#endif
extern "C" int pdm_tune_fps_variant(int v) { const int old = g_fps_variant; g_fps_variant = v; return old; }

#define FPS_LAUNCH(BLOCK, PPT)                                                                  \
    hipLaunchKernelGGL((fps_reg_kernel<BLOCK, PPT>), dim3(b), dim3(BLOCK), 0, as_stream(stream), \
                       n, m, S, logS, points, temp, idx, rg)
#define FPS_PRUNED(BLOCK, PPT, SUB)                                                                                  \
    hipLaunchKernelGGL((fps_pruned_kernel<BLOCK, PPT, SUB, false>), dim3(b), dim3(BLOCK), 0, as_stream(stream), n, m, 1, \
                       points, temp, idx, (unsigned long long *)nullptr, FpsSeg{})
// the shape the 16384-point chunk runs in (single call, resumable jobs and cooperating workgroups alike)
#define FPS_L1_BLOCK 1024
#define FPS_L1_PPT 16
#define FPS_L1_SUB 4

extern "C" int pdm_furthest_point_sampling(void *stream, int b, int n, int m, const float *points,
                                           float *temp, int *idx) {
    PDM_REQUIRE(b >= 0 && n >= 0, PDM_E_BADARG, "fps: negative size b=%d n=%d", b, n);
    if (m <= 0 || b == 0) return 0;  // sampling_gpu.cu:108
    PDM_REQUIRE(n >= 1, PDM_E_BADARG, "fps: n=%d with m=%d", n, m);
    PDM_REQUIRE(points && temp && idx, PDM_E_BADARG, "fps: null pointer");
    int logS = 0;
    const int S = ref_block_threads(n, &logS);
    const int per = (n + S - 1) / S;  // points per reference thread
    const FpsRagged rg{nullptr, nullptr};
    const int small = g_fps_variant >= 16 ? g_fps_variant - 16 : -1;   // 1024 <= n <= 8192 forms
    if (S <= 64) {
        FPS_LAUNCH(64, 2);
    } else if (S <= 256) {
        FPS_LAUNCH(256, 2);
    } else if (S == 512) {
        FPS_LAUNCH(512, 2);
    } else if (per <= 1) {
        // (S = 1024 reference threads played by 256 physical ones: measured 12-15 % faster than 1024 x per for these
        //  sizes, and a quarter of the CU's registers instead of all of them; variant 8 = the 1024-thread form)
        if (g_fps_variant == 8) FPS_LAUNCH(1024, 1);
        else if (small == 1) FPS_PRUNED(256, 4, 1);
        else if (small == 2) FPS_PRUNED(256, 4, 2);
        else FPS_LAUNCH(256, 4);
    } else if (per <= 2) {
        if (g_fps_variant == 8) FPS_LAUNCH(1024, 2);
        else if (small == 1) FPS_PRUNED(256, 8, 1);
        else if (small == 2) FPS_PRUNED(256, 8, 2);
        else if (small == 3) FPS_PRUNED(256, 8, 4);
        else FPS_LAUNCH(256, 8);
    } else if (per <= 4) {
        if (g_fps_variant == 8) FPS_LAUNCH(1024, 4);
        else if (small == 1) FPS_PRUNED(256, 16, 1);
        else if (small == 2) FPS_PRUNED(256, 16, 2);
        else if (small == 3) FPS_PRUNED(256, 16, 4);
        else FPS_LAUNCH(256, 16);
    } else if (per <= 8) {
        if (g_fps_variant == 8) FPS_LAUNCH(1024, 8);
        else if (small == 1) FPS_PRUNED(512, 16, 1);
        else if (small == 2) FPS_PRUNED(512, 16, 2);
        else if (small == 3) FPS_PRUNED(512, 16, 4);
        else if (small == 0) FPS_LAUNCH(512, 16);   // 8192 -> 2048: 2.19 -> 1.73 ms
        else FPS_PRUNED(512, 16, 4);                // 1.73 -> 1.40 ms (lidar-like clouds 1.52)
    } else if (per <= 16) {
        if (g_fps_variant == 1) FPS_LAUNCH(512, 32);
        else if (g_fps_variant == 2) FPS_LAUNCH(1024, 16);
        else if (g_fps_variant == 3) FPS_PRUNED(512, 32, 1);
        else if (g_fps_variant == 4) FPS_PRUNED(1024, 16, 1);
        else if (g_fps_variant == 5) FPS_PRUNED(1024, 16, 2);
        else if (g_fps_variant == 6) FPS_PRUNED(512, 32, 2);
        else if (g_fps_variant == 7) FPS_PRUNED(512, 32, 4);
        else if (g_fps_variant == 9) FPS_PRUNED(512, 32, 8);
        else if (g_fps_variant == 10) FPS_PRUNED(1024, 16, 4);
        else FPS_PRUNED(FPS_L1_BLOCK, FPS_L1_PPT, FPS_L1_SUB);
    } else {
        hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(b), dim3(1024), 0, as_stream(stream), n,
                           m, S, logS, points, temp, idx, rg);
    }
    return check_launch("furthest_point_sampling");
}

// Resumable FPS: job q computes samples [j0[q], j1[q]) of its own batch (b clouds of n points, m samples), reading the
// state of an earlier call from temp[q] (running min-distances; 1e10 everywhere before the first segment) and idx[q]
// (samples [0, j0)), and leaving the state for the next segment there.  The segments of one batch, run in order,
// give exactly the indices of one pdm_furthest_point_sampling call; the jobs of one call belong to DIFFERENT batches
// and run side by side in one launch (pdm_ssd_amd/pipeline.py spreads the long level-1 FPS over several steps).
// Register-resident pruned forms only: 1024 < n <= 16384 (one workgroup per cloud) or, with a workspace of
// pdm_furthest_point_sampling_ws_bytes(b, n) bytes PER JOB, 16384 < n <= 131072 (cooperating workgroups; all of them must be
// co-resident: njobs * b * ceil(n/16384) <= 256).
extern "C" size_t pdm_furthest_point_sampling_ws_bytes(int b, int n);
extern "C" int pdm_fps_max_coresident_workgroups(void);

extern "C" int pdm_furthest_point_sampling_jobs(void *stream, int njobs, int b, int n, int m,
                                                const float *const *points, float *const *temp, int *const *idx,
                                                const int *j0, const int *j1, void *const *workspace,
                                                size_t workspace_bytes) {
    PDM_REQUIRE(njobs >= 1 && njobs <= 4, PDM_E_BADARG, "fps_jobs: njobs=%d not in [1,4]", njobs);
    PDM_REQUIRE(b >= 0 && m >= 1, PDM_E_BADARG, "fps_jobs: b=%d m=%d", b, m);
    const int G = (n + 16383) / 16384;
    PDM_REQUIRE(n > 1024 && G <= 8, PDM_E_BADARG, "fps_jobs: n=%d outside (1024, 131072]", n);
    PDM_REQUIRE(points && temp && idx && j0 && j1, PDM_E_BADARG, "fps_jobs: null table");
    if (b == 0) return 0;
    const size_t ws_need = pdm_furthest_point_sampling_ws_bytes(b, n);
    // the G workgroups of a cloud wait for each other: every workgroup of the launch must be resident at once
    const int cap = G == 1 ? 0 : pdm_fps_max_coresident_workgroups();
    PDM_REQUIRE(G == 1 || (long long)njobs * b * G <= cap, PDM_E_TOOLARGE,
                "fps_jobs: %d jobs x %d clouds x %d workgroups exceed the %d co-resident workgroups of this device", njobs,
                b, G, cap);
    PDM_REQUIRE(G == 1 || (workspace && workspace_bytes >= ws_need), PDM_E_BADARG,
                "fps_jobs: n=%d needs a workspace of %zu bytes per job", n, ws_need);
    FpsSeg seg{};
    seg.njobs = njobs;
    seg.nb = b;
    for (int q = 0; q < njobs; ++q) {
        PDM_REQUIRE(points[q] && temp[q] && idx[q], PDM_E_BADARG, "fps_jobs: null pointer in job %d", q);
        PDM_REQUIRE(j0[q] >= 1 && j0[q] <= j1[q] && j1[q] <= m, PDM_E_BADARG, "fps_jobs: job %d range [%d, %d) of %d", q,
                    j0[q], j1[q], m);
        seg.xyz[q] = points[q]; seg.temp[q] = temp[q]; seg.idx[q] = idx[q];
        seg.j0[q] = j0[q]; seg.j1[q] = j1[q];
        if (G > 1) {
            PDM_REQUIRE(workspace[q] && (reinterpret_cast<uintptr_t>(workspace[q]) & 7) == 0, PDM_E_BADARG,
                        "fps_jobs: workspace of job %d null or not 8-byte aligned", q);
            seg.xch[q] = reinterpret_cast<unsigned long long *>(workspace[q]);
            seg.status[q] = reinterpret_cast<unsigned *>(static_cast<char *>(workspace[q]) + ws_need - 64);
            // stale tags from an earlier segment must not look like iteration numbers of this one
            hipError_t e = hipMemsetAsync(workspace[q], 0, ws_need, as_stream(stream));
            if (e != hipSuccess) {
                set_error("fps_jobs: memset failed: %s", hipGetErrorString(e));
                return (int)e;
            }
        }
    }
    if (G == 1)
        hipLaunchKernelGGL((fps_pruned_kernel<FPS_L1_BLOCK, FPS_L1_PPT, FPS_L1_SUB, false>), dim3(b * njobs), dim3(FPS_L1_BLOCK), 0, as_stream(stream), n, m, 1,
                           (const float *)nullptr, (float *)nullptr, (int *)nullptr, (unsigned long long *)nullptr, seg);
    else
        hipLaunchKernelGGL((fps_pruned_kernel<FPS_L1_BLOCK, FPS_L1_PPT, FPS_L1_SUB, true>), dim3(b * njobs * G), dim3(FPS_L1_BLOCK), 0, as_stream(stream), n, m, G,
                           (const float *)nullptr, (float *)nullptr, (int *)nullptr, (unsigned long long *)nullptr, seg);
    return check_launch("furthest_point_sampling_jobs");
}

// Stacked batches (reference stack_farthest_point_sampling_wrapper, pointnet2_stack/src/sampling.cpp): sample b has
// xyz_batch_cnt[b] points and yields num_sampled_points[b] GLOBAL indices, packed sample after sample.  max_n = the
// largest per-sample point count (host value: it selects the register-resident instantiation).  Tie order is the
// reference's 1024-thread tree for every sample size.  Samples with num_sampled_points > 0 need >= 1 point.
extern "C" int pdm_stack_furthest_point_sampling(void *stream, int B, int max_n, const float *xyz, float *temp,
                                                 const int *xyz_batch_cnt, int *idxs, const int *num_sampled_points) {
    PDM_REQUIRE(B >= 0 && max_n >= 0, PDM_E_BADARG, "stack_fps: B=%d max_n=%d", B, max_n);
    if (B == 0 || max_n == 0) return 0;
    PDM_REQUIRE(xyz && temp && xyz_batch_cnt && idxs && num_sampled_points, PDM_E_BADARG, "stack_fps: null pointer");
    const FpsRagged rg{xyz_batch_cnt, num_sampled_points};
    const int S = 1024, logS = 10, b = B, n = 0, m = 0;
    const float *points = xyz;
    int *idx = idxs;
    const int per = (max_n + 1023) / 1024;
    if (per <= 1) FPS_LAUNCH(1024, 1);
    else if (per <= 2) FPS_LAUNCH(1024, 2);
    else if (per <= 4) FPS_LAUNCH(1024, 4);
    else if (per <= 8) FPS_LAUNCH(1024, 8);
    else if (per <= 16) FPS_LAUNCH(1024, 16);
    else hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(b), dim3(1024), 0, as_stream(stream), n, m, S, logS, points,
                            temp, idx, rg);
    return check_launch("stack_furthest_point_sampling");
}

// Large clouds (n > 16384): G = ceil(n / 16384) cooperating workgroups per cloud, exchange slots in the
// caller's workspace.  Same indices as pdm_furthest_point_sampling.
extern "C" size_t pdm_furthest_point_sampling_ws_bytes(int b, int n) {
    if (b <= 0 || n <= 16384) return 0;
    const int G = (n + 16383) / 16384;
    return (size_t)b * 2 * G * 6 * sizeof(unsigned long long) + 64;   // exchange slots + status word(s)
}

// How many workgroups of the cooperating (multi-workgroup) FPS kernel the current device keeps resident at once:
// compute units x workgroups per unit, asked from the runtime (a partitioned or smaller device gives a smaller
// number).  The G workgroups of a cloud wait for each other, so a launch never holds more than this.
extern "C" int pdm_fps_max_coresident_workgroups(void) {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cached[dev] > 0) return cached[dev];
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &per_cu, fps_pruned_kernel<FPS_L1_BLOCK, FPS_L1_PPT, FPS_L1_SUB, true>, FPS_L1_BLOCK, 0) != hipSuccess)
        return 0;
    cached[dev] = cus * per_cu;
    return cached[dev];
}

// Status of the last cooperating-workgroup FPS that used `workspace` (b clouds of n points; for the jobs form the
// workspace of one job): *flag = 0 ok, 1 = a workgroup gave up waiting for a peer (the indices are not to be used).
// Synchronises `stream`.  Meaningful only for n > 16384.
extern "C" int pdm_furthest_point_sampling_status(void *stream, int b, int n, const void *workspace, int *flag) {
    PDM_REQUIRE(flag, PDM_E_BADARG, "fps_status: null flag");
    *flag = 0;
    const size_t bytes = pdm_furthest_point_sampling_ws_bytes(b, n);
    if (bytes == 0) return 0;
    PDM_REQUIRE(workspace, PDM_E_BADARG, "fps_status: null workspace");
    unsigned words[16];
    hipError_t e = hipMemcpyAsync(words, static_cast<const char *>(workspace) + bytes - 64, sizeof(words),
                                  hipMemcpyDeviceToHost, as_stream(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    if (e != hipSuccess) {
        set_error("fps_status: %s", hipGetErrorString(e));
        return (int)e;
    }
    for (unsigned w : words) *flag |= (w != 0);
    return 0;
}

extern "C" int pdm_furthest_point_sampling_ws(void *stream, int b, int n, int m, const float *points,
                                              float *temp, int *idx, void *workspace, size_t workspace_bytes) {
    const int G = (n + 16383) / 16384;
    if (n <= 16384 || G > 8 || b <= 0 || m <= 0)  // small clouds and clouds beyond 131072 points: single-workgroup kernels
        return pdm_furthest_point_sampling(stream, b, n, m, points, temp, idx);
    PDM_REQUIRE(points && temp && idx && workspace, PDM_E_BADARG, "fps_ws: null pointer");
    PDM_REQUIRE(workspace_bytes >= pdm_furthest_point_sampling_ws_bytes(b, n) &&
                    (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                PDM_E_BADARG, "fps_ws: workspace of %zu bytes, need %zu (8-byte aligned)", workspace_bytes,
                pdm_furthest_point_sampling_ws_bytes(b, n));
    // stale tags from an earlier use of the workspace must not look like iteration numbers
    hipError_t e = hipMemsetAsync(workspace, 0, pdm_furthest_point_sampling_ws_bytes(b, n), as_stream(stream));
    if (e != hipSuccess) {
        set_error("fps_ws: memset failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    // all G workgroups of a cloud must be co-resident (they wait for each other): the launch is cut into chunks of
    // clouds that the device holds at once; a device that cannot hold even one cloud's workgroups takes the
    // single-workgroup streaming kernel
    const int chunk = pdm_fps_max_coresident_workgroups() / G;
    if (chunk < 1) return pdm_furthest_point_sampling(stream, b, n, m, points, temp, idx);
    FpsSeg seg{};
    seg.status[0] = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + pdm_furthest_point_sampling_ws_bytes(b, n) - 64);
    for (int b0 = 0; b0 < b; b0 += chunk) {
        const int nb = b - b0 < chunk ? b - b0 : chunk;
        hipLaunchKernelGGL((fps_pruned_kernel<FPS_L1_BLOCK, FPS_L1_PPT, FPS_L1_SUB, true>), dim3(nb * G), dim3(FPS_L1_BLOCK), 0, as_stream(stream), n, m, G,
                           points + (size_t)b0 * n * 3, temp + (size_t)b0 * n, idx + (size_t)b0 * m,
                           reinterpret_cast<unsigned long long *>(workspace) + (size_t)b0 * 2 * G * 6, seg);
        int rc = check_launch("furthest_point_sampling_ws");
        if (rc) return rc;
    }
    return 0;
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
