// PDM neck, gather form: the same sum as pdm_scatter_bev + pdm_bev_normalize (DESIGN.md "PDM spec"), computed
// per BEV cell instead of per point — no atomics, no zero-fill pass, normalisation fused, every cell of the grid
// written exactly once, and a fixed summation order (ascending point index), so the output is bitwise
// reproducible.  The scatter form runs at the memory-side float-atomic rate (1.26 of ~1.3 TB/s measured) and
// moves ~2.5x the grid through HBM (memset, atomics, normalise); this form moves the grid once.
//
//   bin     (one workgroup per sample): each point is appended to the list of every 8x8-cell tile its dilation
//           block overlaps (count -> scan -> fill; fill order within a list is arbitrary);
//   gather  (one workgroup per (sample, tile)): rank-sorts its list by point index (the fixed summation
//           order), then stages the points in LDS in chunks of 32; phase 1
//           threads evaluate w(cell, point) for the tile's cells, phase 2 waves walk the cells with lanes over
//           channels, acc[cell][ch] += w * f[point][ch] in an LDS accumulator tile; finally acc / wsum is
//           written as channels-last rows (512 contiguous bytes per cell at C = 128).
#include "common.h"

namespace pdm {

constexpr int PG_TS = 8;         // tile edge in cells
#ifndef PG_CHUNK_N
#define PG_CHUNK_N 16
#endif
constexpr int PG_CHUNK = PG_CHUNK_N;     // points staged per pass
constexpr int PG_THREADS = 256;
constexpr int PG_MAXSH = 16;

struct PgGrid {
    float ox, oy, oz, cx, cy, cz, icx, icy, icz;
    int W, H, D, kx, ky, kz, TW, TH;
};

__device__ __forceinline__ int pg_sh_basis(int degree, float x, float y, float z, float *Y) {
    Y[0] = 0.28209479177387814f;
    if (degree < 1) return 1;
    Y[1] = 0.4886025119029199f * y;
    Y[2] = 0.4886025119029199f * z;
    Y[3] = 0.4886025119029199f * x;
    if (degree < 2) return 4;
    const float xx = x * x, yy = y * y, zz = z * z;
    Y[4] = 1.0925484305920792f * (x * y);
    Y[5] = 1.0925484305920792f * (y * z);
    Y[6] = 0.31539156525252005f * (3.0f * zz - 1.0f);
    Y[7] = 1.0925484305920792f * (x * z);
    Y[8] = 0.5462742152960396f * (xx - yy);
    if (degree < 3) return 9;
    Y[9] = 0.5900435899266435f * (y * (3.0f * xx - yy));
    Y[10] = 2.890611442640554f * (x * y * z);
    Y[11] = 0.4570457994644658f * (y * (5.0f * zz - 1.0f));
    Y[12] = 0.3731763325901154f * (z * (5.0f * zz - 3.0f));
    Y[13] = 0.4570457994644658f * (x * (5.0f * zz - 1.0f));
    Y[14] = 1.445305721320277f * (z * (xx - yy));
    Y[15] = 0.5900435899266435f * (x * (xx - 3.0f * yy));
    return 16;
}

__device__ __forceinline__ bool pg_base_cell(const PgGrid &g, float px, float py, float pz, int &bx, int &by, int &bz) {
    if (!(px == px) || !(py == py) || !(pz == pz)) return false;
    const float fx = floorf(__fmul_rn(px - g.ox, g.icx));
    const float fy = floorf(__fmul_rn(py - g.oy, g.icy));
    const float fz = floorf(__fmul_rn(pz - g.oz, g.icz));
    if (fx < -(float)g.kx || fx > (float)(g.W + g.kx) || fy < -(float)g.ky || fy > (float)(g.H + g.ky) ||
        fz < -(float)g.kz || fz > (float)(g.D + g.kz))
        return false;
    bx = (int)fx; by = (int)fy; bz = (int)fz;
    return true;
}

// tile range [t0, t1] (may be empty) covered by cells [c - h, c + h] clipped to [0, n)
__device__ __forceinline__ void pg_tile_range(int c, int h, int ncell, int ntile, int &t0, int &t1) {
    const int lo = max(c - h, 0), hi = min(c + h, ncell - 1);
    if (lo > hi) { t0 = 1; t1 = 0; return; }
    t0 = lo / PG_TS; t1 = min(hi / PG_TS, ntile - 1);
}

// tile_start (B, ntiles + 1), tile_pts (B, cap)
__global__ __launch_bounds__(PG_THREADS) void pdm_bin_kernel(int P, PgGrid g, int ntiles, int cap,
                                                             const float *__restrict__ xyz,
                                                             int *__restrict__ tile_start_all,
                                                             int *__restrict__ tile_pts_all) {
    extern __shared__ int cnt[];  // ntiles counters, then reused as cursors
    __shared__ int wsum[PG_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *__restrict__ tile_start = tile_start_all + (size_t)b * (ntiles + 1);
    int *__restrict__ tile_pts = tile_pts_all + (size_t)b * cap;
    for (int t = tid; t < ntiles; t += PG_THREADS) cnt[t] = 0;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = tid; i < P; i += PG_THREADS) {
            const float *p = xyz + ((size_t)b * P + i) * 3;
            int bx, by, bz;
            if (!pg_base_cell(g, p[0], p[1], p[2], bx, by, bz)) continue;
            if (bz + g.kz / 2 < 0 || bz - g.kz / 2 >= g.D) continue;
            int tx0, tx1, ty0, ty1;
            pg_tile_range(bx, g.kx / 2, g.W, g.TW, tx0, tx1);
            pg_tile_range(by, g.ky / 2, g.H, g.TH, ty0, ty1);
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) {
                    const int slot = atomicAdd(&cnt[ty * g.TW + tx], 1);
                    if (pass == 1) tile_pts[slot] = i;
                }
        }
        __syncthreads();
        if (pass == 0) {
            // exclusive scan of cnt -> tile_start, cnt becomes the running cursor
            const int per = (ntiles + PG_THREADS - 1) / PG_THREADS;
            const int c0 = tid * per, c1 = min(c0 + per, ntiles);
            int local = 0;
            for (int t = c0; t < c1; ++t) local += cnt[t];
            int incl = local;
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; ++w) base += wsum[w];
            int run = base + incl - local;
            for (int t = c0; t < c1; ++t) {
                const int v = cnt[t];
                cnt[t] = run;
                tile_start[t] = run;
                run += v;
            }
            if (tid == PG_THREADS - 1) tile_start[ntiles] = run;
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(PG_THREADS) void pdm_gather_kernel(
    int P, int C, int degree, PgGrid g, int ntiles, int cap, int normalize, float eps,
    const float *__restrict__ xyz, const float *__restrict__ feat, const float *__restrict__ sh,
    const float *__restrict__ inv2s2, const int *__restrict__ tile_start_all, const int *__restrict__ tile_pts_all,
    int *__restrict__ tile_sorted_all, float *__restrict__ grid, float *__restrict__ wsum_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ncell = PG_TS * PG_TS * g.D;                    // cells of the tile, index = (yy*8 + xx)*D + z
    float *acc = lds;                                         // ncell x C
    float *wsm = acc + (size_t)ncell * C;                     // ncell
    float *wgt = wsm + ncell;                                 // ncell x PG_CHUNK
    float *fbuf = wgt + (size_t)ncell * PG_CHUNK;             // PG_CHUNK x C
    float *par = fbuf + (size_t)PG_CHUNK * C;                 // PG_CHUNK x (4 + PG_MAXSH) : x,y,z,inv2s2, sh...
    int *pcell = reinterpret_cast<int *>(par + PG_CHUNK * (4 + PG_MAXSH));  // PG_CHUNK x 3 base cells
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile % g.TW, ty = tile / g.TW;
    const int nsh = (degree + 1) * (degree + 1);
    const int *__restrict__ ts = tile_start_all + (size_t)b * (ntiles + 1);
    const int s = ts[tile], e = ts[tile + 1];
    int *tp = tile_sorted_all + (size_t)b * cap;
    {   // point indices are distinct, so rank = number of smaller entries; O(n^2 / 256), n is a handful
        const int *__restrict__ raw = tile_pts_all + (size_t)b * cap;
        for (int i = s + tid; i < e; i += PG_THREADS) {
            const int v = raw[i];
            int rank = 0;
            for (int j = s; j < e; ++j) rank += raw[j] < v;
            tp[s + rank] = v;
        }
    }

    for (int i = tid; i < ncell * C; i += PG_THREADS) acc[i] = 0.0f;
    for (int i = tid; i < ncell; i += PG_THREADS) wsm[i] = 0.0f;
    for (int c0 = s; c0 < e; c0 += PG_CHUNK) {
        const int np = min(PG_CHUNK, e - c0);
        __syncthreads();
        // stage the chunk's points
        for (int i = tid; i < np * C; i += PG_THREADS) {
            const int q = i / C, c = i - q * C;
            fbuf[q * C + c] = feat[((size_t)b * P + tp[c0 + q]) * C + c];
        }
        for (int i = tid; i < np * (4 + PG_MAXSH); i += PG_THREADS) {
            const int q = i / (4 + PG_MAXSH), k = i - q * (4 + PG_MAXSH);
            const size_t pi = (size_t)b * P + tp[c0 + q];
            float v = 0.0f;
            if (k < 3) v = xyz[pi * 3 + k];
            else if (k == 3) v = inv2s2[pi];
            else if (k - 4 < nsh) v = sh[pi * nsh + (k - 4)];
            par[i] = v;
        }
        __syncthreads();
        if (tid < np) {
            int bx = 0, by = 0, bz = 0;
            pg_base_cell(g, par[tid * (4 + PG_MAXSH)], par[tid * (4 + PG_MAXSH) + 1], par[tid * (4 + PG_MAXSH) + 2], bx, by, bz);
            pcell[tid * 3] = bx; pcell[tid * 3 + 1] = by; pcell[tid * 3 + 2] = bz;
        }
        __syncthreads();
        // phase 1: weights of (cell, point) pairs
        for (int i = tid; i < ncell * np; i += PG_THREADS) {
            const int cl = i / np, q = i - cl * np;
            const int z = cl % g.D, xy = cl / g.D;
            const int gx = tx * PG_TS + (xy % PG_TS), gy = ty * PG_TS + (xy / PG_TS);
            float w = 0.0f;
            const int bx = pcell[q * 3], by = pcell[q * 3 + 1], bz = pcell[q * 3 + 2];
            if (gx < g.W && gy < g.H && abs(gx - bx) <= g.kx / 2 && abs(gy - by) <= g.ky / 2 && abs(z - bz) <= g.kz / 2) {
                const float *pp = par + q * (4 + PG_MAXSH);
                const float ux = __fmaf_rn((float)gx + 0.5f, g.cx, g.ox) - pp[0];
                const float uy = __fmaf_rn((float)gy + 0.5f, g.cy, g.oy) - pp[1];
                const float uz = __fmaf_rn((float)z + 0.5f, g.cz, g.oz) - pp[2];
                const float r2 = sqdist(ux, uy, uz);
                float sacc;
                if (r2 > 0.0f) {
                    const float inv = 1.0f / sqrtf(r2);
                    float Y[PG_MAXSH];
                    const int ny = pg_sh_basis(degree, ux * inv, uy * inv, uz * inv, Y);
                    sacc = 0.0f;
#pragma unroll
                    for (int t = 0; t < PG_MAXSH; ++t)
                        if (t < ny) sacc = __fmaf_rn(pp[4 + t], Y[t], sacc);
                } else {
                    sacc = pp[4] * 0.28209479177387814f;
                }
                w = sacc * __expf(-r2 * pp[3]);
            }
            wgt[cl * PG_CHUNK + q] = w;
        }
        __syncthreads();
        // phase 2: waves over cells, lanes over channels
        for (int cl = wave; cl < ncell; cl += PG_THREADS / 64) {
            const float *wr = wgt + cl * PG_CHUNK;
            float ws = 0.0f;
            for (int q = 0; q < np; ++q) ws += wr[q];
            if (lane == 0) wsm[cl] += ws;
            for (int c = lane; c < C; c += 64) {
                float a = acc[cl * C + c];
                for (int q = 0; q < np; ++q) {
                    const float w = wr[q];
                    if (w != 0.0f) a += w * fbuf[q * C + c];  // wave-uniform branch
                }
                acc[cl * C + c] = a;
            }
        }
    }
    __syncthreads();
    // write the tile: grid (B,H,W,C*D) with inner index c*D + z, wsum (B,H,W,D)
    const int CD = C * g.D;
    for (int xy = wave; xy < PG_TS * PG_TS; xy += PG_THREADS / 64) {
        const int gx = tx * PG_TS + (xy % PG_TS), gy = ty * PG_TS + (xy / PG_TS);
        if (gx >= g.W || gy >= g.H) continue;
        float *dst = grid + (((size_t)b * g.H + gy) * g.W + gx) * CD;
        for (int q = lane; q < CD; q += 64) {
            const int c = q / g.D, z = q - c * g.D;
            const int cl = xy * g.D + z;
            float v = acc[cl * C + c];
            const float ws = wsm[cl];
            if (normalize && fabsf(ws) > eps) v *= 1.0f / ws;
            dst[q] = v;
        }
        if (lane < g.D) wsum_out[(((size_t)b * g.H + gy) * g.W + gx) * g.D + lane] = wsm[xy * g.D + lane];
    }
}


// D == 1 form: the accumulator tile lives in registers.  Wave w owns tile rows 2w and 2w+1 (16 cells), lane l
// owns channels l, l+64, ... (CPL per lane), so acc[16][CPL] never leaves the register file, an empty tile costs
// nothing but its zero stores, and every store instruction of the epilogue is one 256-byte row segment.
//
// A tile of the bench workload holds a handful of points (1024 points x ~3 tiles each over 550 tiles), so a workgroup
// is a chain of dependent round trips (list bounds -> list -> point rows -> stores) rather than arithmetic:
//   - the list is rank-sorted from LDS into LDS (no store + reload of a sorted copy through L2);
//   - a point reaches ~16 of the tile's 64 cells: phase 1 leaves the ballot of its non-zero weights beside them, and a
//     wave skips a point that misses its two rows (and the 4-cell groups it misses) on scalar tests — adding a zero
//     weight changes nothing, so the sums are those of the dense loop, bit for bit;
//   - the per-cell weight sums are taken by 16 lanes of the wave from the same LDS rows, in the same order.
#ifndef PG_LIST_N
#define PG_LIST_N 256
#endif
constexpr int PG_LIST = PG_LIST_N;    // list entries sorted in LDS; a longer list (never at the shipped sizes) sorts through global memory

#ifndef PG_DIAG
#define PG_DIAG 0                // timing builds (tools/diag/pdm_gather_rate.py): 1 no accumulation, 2 no weight arithmetic, 4 stores only, 8 no list (points tile*7.. of the sample)
#endif

typedef float pg_v2f __attribute__((ext_vector_type(2)));

// 16-point stages, a 256-entry LDS list and 5 waves per SIMD for C <= 128 (96 VGPRs, 2 spills): five workgroups per CU instead of
// four, 224 -> 212 us per call on the bench workload (same box, profiles/r03b_pdm_gather_diag.txt); six or more spill heavily.
template <int CPL>
__global__ __launch_bounds__(PG_THREADS, CPL <= 2 ? 5 : 1) void pdm_gather_reg_kernel(
    int P, int C, int degree, PgGrid g, int ntiles, int cap, int normalize, float eps,
    const float *__restrict__ xyz, const float *__restrict__ feat, const float *__restrict__ sh,
    const float *__restrict__ inv2s2, const int *__restrict__ tile_start_all, const int *__restrict__ tile_pts_all,
    int *__restrict__ tile_sorted_all, float *__restrict__ grid, float *__restrict__ wsum_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NPAR = 4 + PG_MAXSH;
    constexpr int CP2 = (CPL + 1) / 2;             // channel pairs per lane: (k, k + 1) of a pair are channels 64 k + lane, 64 (k + 1) + lane
    float *wgt = lds;                              // PG_CHUNK x 64 cells
    float *fbuf = wgt + PG_CHUNK * 64;             // PG_CHUNK x C
    float *par = fbuf + (size_t)PG_CHUNK * C;      // PG_CHUNK x NPAR : x, y, z, inv2s2, sh...
    int *pbase = reinterpret_cast<int *>(par + PG_CHUNK * NPAR);                 // PG_CHUNK x 2: base cell (x, y), or a cell no tile reaches
    unsigned *msk = reinterpret_cast<unsigned *>(pbase + PG_CHUNK * 2);          // PG_CHUNK x 2: ballot of the non-zero weights
    int *lst = reinterpret_cast<int *>(msk + PG_CHUNK * 2);                      // PG_LIST unsorted, PG_LIST sorted
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = tile % g.TW, ty = tile / g.TW;
    const int nsh = (degree + 1) * (degree + 1);
    const int *__restrict__ ts = tile_start_all + (size_t)b * (ntiles + 1);
    const int s = PG_DIAG & 4 ? 0 : ts[tile], e = PG_DIAG & 4 ? 0 : ts[tile + 1];
    const int *sorted = lst + PG_LIST;
    pg_v2f acc[16][CP2];
    float wsl = 0.0f;                              // lanes 0..15: weight sum of cell wave * 16 + lane
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int k = 0; k < CP2; ++k) acc[j][k] = pg_v2f{0.0f, 0.0f};
    if (e > s && !(PG_DIAG & 8)) {   // distinct indices: rank = number of smaller entries
        const int *__restrict__ raw = tile_pts_all + (size_t)b * cap;
        if (e - s <= PG_LIST) {
            for (int i = tid; i < e - s; i += PG_THREADS) lst[i] = raw[s + i];
            __syncthreads();
            for (int i = tid; i < e - s; i += PG_THREADS) {
                const int v = lst[i];
                int rank = 0;
                for (int j = 0; j < e - s; ++j) rank += lst[j] < v;
                lst[PG_LIST + rank] = v;
            }
        } else {
            int *tp = tile_sorted_all + (size_t)b * cap;
            for (int i = s + tid; i < e; i += PG_THREADS) {
                const int v = raw[i];
                int rank = 0;
                for (int j = s; j < e; ++j) rank += raw[j] < v;
                tp[s + rank] = v;
            }
            sorted = nullptr;
        }
    }
    const int *tpg = tile_sorted_all + (size_t)b * cap + s;
    for (int c0 = 0; c0 < e - s; c0 += PG_CHUNK) {
        const int np = min(PG_CHUNK, e - s - c0);
        __syncthreads();
#define PG_POINT(q) (PG_DIAG & 8 ? (tile * 7 + c0 + (q)) % P : sorted ? sorted[c0 + (q)] : tpg[c0 + (q)])
        if ((C & 3) == 0) {
            const int C4 = C >> 2;
            for (int i = tid; i < np * C4; i += PG_THREADS) {
                const int q = i / C4, c = i - q * C4;
                reinterpret_cast<float4 *>(fbuf)[q * C4 + c] =
                    reinterpret_cast<const float4 *>(feat + ((size_t)b * P + PG_POINT(q)) * C)[c];
            }
        } else {
            for (int i = tid; i < np * C; i += PG_THREADS) {
                const int q = i / C, c = i - q * C;
                fbuf[q * C + c] = feat[((size_t)b * P + PG_POINT(q)) * C + c];
            }
        }
        for (int i = tid; i < np * PG_MAXSH; i += PG_THREADS) {
            const int q = i / PG_MAXSH, k = i - q * PG_MAXSH;
            par[q * NPAR + 4 + k] = k < nsh ? sh[((size_t)b * P + PG_POINT(q)) * nsh + k] : 0.0f;
        }
        if (tid < np) {
            const size_t pi = (size_t)b * P + PG_POINT(tid);
            const float px = xyz[pi * 3], py = xyz[pi * 3 + 1], pz = xyz[pi * 3 + 2];
            par[tid * NPAR] = px; par[tid * NPAR + 1] = py; par[tid * NPAR + 2] = pz; par[tid * NPAR + 3] = inv2s2[pi];
            int bx = 0, by = 0, bz = 0;
            const bool in = pg_base_cell(g, px, py, pz, bx, by, bz) && abs(bz) <= g.kz / 2;
            pbase[tid * 2] = in ? bx : -(1 << 28);
            pbase[tid * 2 + 1] = by;
        }
#undef PG_POINT
        __syncthreads();
        for (int q = wave; q < np; q += PG_THREADS / 64) {   // weight of (point q, cell `lane`): one point per wave pass
            const float *pp = par + q * NPAR;
            const int gx = tx * PG_TS + (lane & 7), gy = ty * PG_TS + (lane >> 3);
            const int bx = pbase[q * 2], by = pbase[q * 2 + 1];
            float w = 0.0f;
            if (!(PG_DIAG & 2) && gx < g.W && gy < g.H && abs(gx - bx) <= g.kx / 2 && abs(gy - by) <= g.ky / 2) {
                const float ux = __fmaf_rn((float)gx + 0.5f, g.cx, g.ox) - pp[0];
                const float uy = __fmaf_rn((float)gy + 0.5f, g.cy, g.oy) - pp[1];
                const float uz = __fmaf_rn(0.5f, g.cz, g.oz) - pp[2];
                const float r2 = sqdist(ux, uy, uz);
                float sacc;
                if (r2 > 0.0f) {
                    const float inv = 1.0f / sqrtf(r2);
                    float Y[PG_MAXSH];
                    const int ny = pg_sh_basis(degree, ux * inv, uy * inv, uz * inv, Y);
                    sacc = 0.0f;
#pragma unroll
                    for (int t = 0; t < PG_MAXSH; ++t)
                        if (t < ny) sacc = __fmaf_rn(pp[4 + t], Y[t], sacc);
                } else {
                    sacc = pp[4] * 0.28209479177387814f;
                }
                w = sacc * __expf(-r2 * pp[3]);
            }
            wgt[q * 64 + lane] = w;
            const unsigned long long m = __ballot(w != 0.0f);
            if (lane == 0) { msk[q * 2] = (unsigned)m; msk[q * 2 + 1] = (unsigned)(m >> 32); }
        }
        __syncthreads();
        if (PG_DIAG & 1) continue;
        for (int q = 0; q < np; ++q) {
            const unsigned m16 = (__builtin_amdgcn_readfirstlane(msk[q * 2 + (wave >> 1)]) >> ((wave & 1) * 16)) & 0xffffu;
            if (m16 == 0) continue;                                     // the point misses this wave's two rows
            wsl += wgt[q * 64 + wave * 16 + (lane & 15)];
            pg_v2f f[CP2];
#pragma unroll
            for (int k = 0; k < CP2; ++k) {
                f[k].x = (2 * k * 64 + lane < C) ? fbuf[q * C + 2 * k * 64 + lane] : 0.0f;
                f[k].y = (2 * k + 1 < CPL && (2 * k + 1) * 64 + lane < C) ? fbuf[q * C + (2 * k + 1) * 64 + lane] : 0.0f;
            }
            const float4 *wr = reinterpret_cast<const float4 *>(wgt + q * 64 + wave * 16);
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                if (!(m16 & (0xfu << (4 * j4)))) continue;
                const float4 w4 = wr[j4];
                const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int k = 0; k < CP2; ++k)
                        acc[j4 * 4 + jj][k] = __builtin_elementwise_fma(pg_v2f{w[jj], w[jj]}, f[k], acc[j4 * 4 + jj][k]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int gx = tx * PG_TS + (j & 7), gy = ty * PG_TS + wave * 2 + (j >> 3);
        const float ws = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wsl), j));
        if (gx >= g.W || gy >= g.H) continue;  // wave-uniform
        const size_t cell = ((size_t)b * g.H + gy) * g.W + gx;
        const float sc = (normalize && fabsf(ws) > eps) ? 1.0f / ws : 1.0f;
        float *dst = grid + cell * C;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
            if (k * 64 + lane < C) dst[k * 64 + lane] = (k & 1 ? acc[j][k >> 1].y : acc[j][k >> 1].x) * sc;
        if (lane == 0) wsum_out[cell] = ws;
    }
}

}  // namespace pdm

using namespace pdm;

static int pg_span(int k) { return (k / 2 * 2 + PG_TS - 1) / PG_TS + 1; }  // tiles a dilation block can touch per axis

extern "C" size_t pdm_gather_bev_workspace_bytes(int B, int P, int W, int H, int kx, int ky) {
    // P = 0 still needs the per-tile list starts (zeroed; the gather kernel reads them to write empty tiles).  (Until round 4 this returned
    // 0 for P = 0: the caller's 16-byte stand-in was then overrun by B (ntiles + 1) ints — unnoticed inside the allocator's 2 MB segment
    // until the block happened to be the last of one, where hipMemsetAsync refused the range.)
    if (B <= 0 || P < 0 || W <= 0 || H <= 0) return 0;
    const long long ntiles = (long long)((W + PG_TS - 1) / PG_TS) * ((H + PG_TS - 1) / PG_TS);
    const long long cap = (long long)P * pg_span(kx) * pg_span(ky);
    return (size_t)B * ((size_t)(ntiles + 1) + 2 * (size_t)cap) * sizeof(int) + 64;
}

extern "C" int pdm_gather_bev(void *stream, int B, int P, int C, int degree, const float *xyz,
                              const float *feat, const float *sh, const float *inv2s2, float ox, float oy,
                              float oz, float cx, float cy, float cz, float icx, float icy, float icz, int W,
                              int H, int D, int kx, int ky, int kz, int normalize, float eps, float *grid,
                              float *wsum, void *workspace, size_t workspace_bytes) {
    PDM_REQUIRE(B >= 0 && P >= 0 && C >= 1, PDM_E_BADARG, "pdm_gather_bev: bad size");
    PDM_REQUIRE(degree >= 0 && degree <= 3, PDM_E_BADARG, "pdm_gather_bev: SH degree %d", degree);
    PDM_REQUIRE(W > 0 && H > 0 && D > 0 && kx > 0 && ky > 0 && kz > 0 && (kx & 1) && (ky & 1) && (kz & 1), PDM_E_BADARG,
                "pdm_gather_bev: grid %dx%dx%d dilation %dx%dx%d", W, H, D, kx, ky, kz);
    if (B == 0) return 0;
    PDM_REQUIRE(grid && wsum && workspace && (P == 0 || (xyz && feat && sh && inv2s2)), PDM_E_BADARG, "pdm_gather_bev: null pointer");
    PgGrid g{ox, oy, oz, cx, cy, cz, icx, icy, icz, W, H, D, kx, ky, kz, (W + PG_TS - 1) / PG_TS, (H + PG_TS - 1) / PG_TS};
    const int ntiles = g.TW * g.TH;
    const long long cap = (long long)P * pg_span(kx) * pg_span(ky);
    PDM_REQUIRE(ntiles <= 16000 && cap < (1ll << 30), PDM_E_TOOLARGE, "pdm_gather_bev: %d tiles / list capacity %lld", ntiles, cap);
    PDM_REQUIRE(workspace_bytes >= pdm_gather_bev_workspace_bytes(B, P, W, H, kx, ky), PDM_E_BADARG,
                "pdm_gather_bev: workspace of %zu bytes, need %zu", workspace_bytes,
                pdm_gather_bev_workspace_bytes(B, P, W, H, kx, ky));
    int *tile_start = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15);
    int *tile_pts = tile_start + (size_t)B * (ntiles + 1);
    int *tile_sorted = tile_pts + (size_t)B * cap;
    const bool reg_form = D == 1 && C <= 256;
    const int ncell = PG_TS * PG_TS * D;
    const size_t lds = reg_form ? ((size_t)PG_CHUNK * 64 + (size_t)PG_CHUNK * C + PG_CHUNK * (4 + PG_MAXSH) + PG_CHUNK * 4 + 2 * PG_LIST) * sizeof(float)
                                : ((size_t)ncell * C + ncell + (size_t)ncell * PG_CHUNK + (size_t)PG_CHUNK * C +
                                   PG_CHUNK * (4 + PG_MAXSH)) * sizeof(float) + PG_CHUNK * 3 * sizeof(int);
    PDM_REQUIRE(lds <= 64 * 1024, PDM_E_TOOLARGE, "pdm_gather_bev: C=%d D=%d need %zu bytes of LDS (use pdm_scatter_bev)", C, D, lds);
    if (P > 0)
        hipLaunchKernelGGL(pdm_bin_kernel, dim3(B), dim3(PG_THREADS), (size_t)ntiles * sizeof(int), as_stream(stream), P, g,
                           ntiles, (int)cap, xyz, tile_start, tile_pts);
    else {
        hipError_t e = hipMemsetAsync(tile_start, 0, (size_t)B * (ntiles + 1) * sizeof(int), as_stream(stream));
        if (e != hipSuccess) {
            set_error("pdm_gather_bev: memset of %zu bytes at %p failed: %s", (size_t)B * (ntiles + 1) * sizeof(int), (void *)tile_start, hipGetErrorString(e));
            (void)hipGetLastError();      // (not to be reported again by the next entry point's launch check)
            return (int)e;
        }
    }
    int rc = check_launch("pdm_gather_bev(bin)");
    if (rc) return rc;
#define PG_LAUNCH(K) hipLaunchKernelGGL(K, dim3(ntiles, B), dim3(PG_THREADS), lds, as_stream(stream), P, C, degree, g, ntiles, \
                                        (int)cap, normalize, eps, xyz, feat, sh, inv2s2, tile_start, tile_pts, tile_sorted, grid, wsum)
    if (!reg_form) PG_LAUNCH(pdm_gather_kernel);
    else if (C <= 64) PG_LAUNCH(pdm_gather_reg_kernel<1>);
    else if (C <= 128) PG_LAUNCH(pdm_gather_reg_kernel<2>);
    else if (C <= 192) PG_LAUNCH(pdm_gather_reg_kernel<3>);
    else PG_LAUNCH(pdm_gather_reg_kernel<4>);
#undef PG_LAUNCH
    return check_launch("pdm_gather_bev(gather)");
}
