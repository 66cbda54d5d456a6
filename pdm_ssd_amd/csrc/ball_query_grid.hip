// Grid-accelerated ball query for gfx950 — same result, bit for bit, as the exhaustive scan of
// ball_query.hip / the reference kernel (ball_query_gpu.cu:15-51): per centre the first `nsample`
// indices in ASCENDING index order with d2 < r^2, padded with the first hit.
//
// Why: the scan is N*M distance evaluations (SA1 at bs=32: 2.1 G per radius) while the answer depends
// on the few points near each centre.  A uniform grid over the cloud bounds the candidates; the
// "first nsample by index" rule is then restored exactly and independently of visiting order:
//   build  (one 1024-thread workgroup per sample): bounding box -> cell size h >= 2r (grown until the
//          grid has <= BQG_CAP cells) -> LDS histogram -> scan -> points scattered into cell order as
//          float4 {x, y, z, index}, so the query streams candidates with coalesced 16-byte loads;
//   query  (one wave64 per centre): the cells overlapping [c - R, c + R] (R = r plus a rounding margin)
//          form at most a few x-contiguous runs; lanes test one candidate each, a hit sets bit `index`
//          in a wave-private LDS bitmap; the bitmap is then read back in index order (per-lane popcount,
//          wave prefix sum) and the first nsample set bits are written.
// Exactness argument (DESIGN.md "ball query"): the cell function is monotone in the coordinate, and
// every point that passes the fp32 test lies in [fl(c - R), fl(c + R)] per axis, so it is in a visited
// cell whatever h is; the distance arithmetic is the same pinned sequence as everywhere else.
#include "grid.h"

namespace pdm {

__device__ __forceinline__ float shfl_xor_f(float v, int m) { return __shfl_xor(v, m, 64); }

// workspace layout per call: [B headers][B * (CAP+1) cell starts][B * n float4 sorted points]
// CACHED (n <= 16 * 1024): a thread keeps its <= 16 points in registers from the bounding-box pass on — the histogram
// pass(es) and the scatter pass re-read nothing (round 2 read the cloud three times through strided 4-byte loads; the
// build is one workgroup per cloud, i.e. a latency chain, and was 30 us of the 100 us of an SA level's neighbour search).
constexpr int BQG_PPT = 16;
// SPLIT (round 4): the kernel stops behind the scan and leaves a second copy of the cell starts (`cursor_all`) for
// bq_grid_scatter_kernel, which places the points from MANY workgroups per cloud: the scatter is 16 stores of 16 bytes
// per thread to 64 different lines per wave-instruction — 8.5 of the 28 us of a 16384-point build when one compute
// unit issues all of them (s_memtime phase table, profiles/r04c_grid_build_phase.txt).
template <bool CACHED, bool SPLIT = false>
__global__ __launch_bounds__(BQG_BUILD_T) void bq_grid_build_kernel(int n, float radius, int max_cells,
                                                                    const float *__restrict__ xyz_all,
                                                                    float *__restrict__ hdr_all,
                                                                    int *__restrict__ cell_start_all,
                                                                    float4 *__restrict__ sorted_all,
                                                                    int *__restrict__ cursor_all = nullptr, int want_density = 0) {
    __shared__ int hist[BQG_CAP];
    __shared__ float red[6][BQG_BUILD_T / 64];
    __shared__ int wsum[BQG_BUILD_T / 64];
    __shared__ GridHdr sh;
    __shared__ float s_h, s_ext[3];
    __shared__ int s_nonempty, s_again;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *__restrict__ xyz = xyz_all + (size_t)b * n * 3;
    int *__restrict__ cell_start = cell_start_all + (size_t)b * (BQG_CAP + 1);
    float4 *__restrict__ sorted = sorted_all + (size_t)b * n;
#ifdef BQG_DIAG   // phase stamps (shader cycles since the start) in the unused header floats 8..15: tools/diag/grid_build_phase.py
    const long long t_start = __builtin_amdgcn_s_memtime();
#define BQG_STAMP(i) do { __syncthreads(); if (tid == 0) hdr_all[(size_t)b * BQG_HDR + 9 + (i)] = (float)(__builtin_amdgcn_s_memtime() - t_start); } while (0)
#else
#define BQG_STAMP(i) ((void)0)
#endif

    // ---- bounding box
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    float px[CACHED ? BQG_PPT : 1], py[CACHED ? BQG_PPT : 1], pz[CACHED ? BQG_PPT : 1];
    if constexpr (CACHED) {
#pragma unroll
        for (int i = 0; i < BQG_PPT; ++i) {
            const int k = tid + i * BQG_BUILD_T;
            px[i] = py[i] = pz[i] = 0.0f;
            if (k < n) {
                px[i] = xyz[(size_t)k * 3 + 0]; py[i] = xyz[(size_t)k * 3 + 1]; pz[i] = xyz[(size_t)k * 3 + 2];
                mn[0] = fminf(mn[0], px[i]); mx[0] = fmaxf(mx[0], px[i]);
                mn[1] = fminf(mn[1], py[i]); mx[1] = fmaxf(mx[1], py[i]);
                mn[2] = fminf(mn[2], pz[i]); mx[2] = fmaxf(mx[2], pz[i]);
            }
        }
    } else {
        for (int k = tid; k < n; k += BQG_BUILD_T) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = xyz[(size_t)k * 3 + a];
                mn[a] = fminf(mn[a], v);
                mx[a] = fmaxf(mx[a], v);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off >= 1; off >>= 1) {
            mn[a] = fminf(mn[a], shfl_xor_f(mn[a], off));
            mx[a] = fmaxf(mx[a], shfl_xor_f(mx[a], off));
        }
        if (lane == 0) { red[a][wave] = mn[a]; red[3 + a][wave] = mx[a]; }
    }
    __syncthreads();
    if (tid == 0) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = red[a][0]; hi[a] = red[3 + a][0];
            for (int w = 1; w < BQG_BUILD_T / 64; ++w) { lo[a] = fminf(lo[a], red[a][w]); hi[a] = fmaxf(hi[a], red[3 + a][w]); }
            if (!(lo[a] <= hi[a])) { lo[a] = 0.0f; hi[a] = 0.0f; }  // empty / all-NaN axis
        }
        // start from the larger of 2r and the finest grid the index range allows (<= 1024 cells per axis),
        // then coarsen by 2x in volume per step until the cell budget is met
        const float max_ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        float h = fmaxf(fmaxf(2.02f * fabsf(radius), max_ext * (1.0f / 1024.0f)), 1e-30f);
        if (!(h == h) || h > 3.0e38f) h = 3.0e38f;
        int g[3];
        for (int it = 0; it < 200; ++it) {
            long long total = 1;
            for (int a = 0; a < 3; ++a) {
                const float q = (hi[a] - lo[a]) / h;
                g[a] = q < 1023.0f ? (int)q + 1 : 1024;   // NaN/inf extents fall to 1024 then shrink
                if (!(q == q)) g[a] = 1;
                total *= g[a];
            }
            if (total <= max_cells) break;
            h *= 1.26f;
        }
        if ((long long)g[0] * g[1] * g[2] > max_cells) { g[0] = g[1] = g[2] = 1; }
        sh.minx = lo[0]; sh.miny = lo[1]; sh.minz = lo[2];
        sh.inv_h = 1.0f / h;
        sh.gx = g[0]; sh.gy = g[1]; sh.gz = g[2];
        sh.ncells = g[0] * g[1] * g[2];
        s_h = h;
        s_ext[0] = hi[0] - lo[0]; s_ext[1] = hi[1] - lo[1]; s_ext[2] = hi[2] - lo[2];
    }
    __syncthreads();
    GridHdr H = sh;
    BQG_STAMP(0);   // loads + bounding box + cell size

    // ---- histogram.  Nearest-neighbour mode (radius == 0: the cell size is free): a cell budget of n/2 assumes the
    // points fill their bounding box; a flat scene (LiDAR: a few metres of height over 70 x 80 m) leaves most cells
    // empty and packs the rest, and every box scan of the query pays for it.  While the occupied cells hold more than
    // two points on average and a grid of half the cell volume still fits the LDS histogram, refine and count again.
    for (int pass = 0;; ++pass) {
        for (int c = tid; c < H.ncells; c += BQG_BUILD_T) hist[c] = 0;
        if (tid == 0) s_nonempty = 0;
        __syncthreads();
        if constexpr (CACHED) {
#pragma unroll
            for (int i = 0; i < BQG_PPT; ++i) {
                if (tid + i * BQG_BUILD_T < n) {
                    const int cx = cell_of(px[i], H.minx, H.inv_h, H.gx);
                    const int cy = cell_of(py[i], H.miny, H.inv_h, H.gy);
                    const int cz = cell_of(pz[i], H.minz, H.inv_h, H.gz);
                    atomicAdd(&hist[(cz * H.gy + cy) * H.gx + cx], 1);
                }
            }
        } else {
            for (int k = tid; k < n; k += BQG_BUILD_T) {
                const int cx = cell_of(xyz[(size_t)k * 3 + 0], H.minx, H.inv_h, H.gx);
                const int cy = cell_of(xyz[(size_t)k * 3 + 1], H.miny, H.inv_h, H.gy);
                const int cz = cell_of(xyz[(size_t)k * 3 + 2], H.minz, H.inv_h, H.gz);
                atomicAdd(&hist[(cz * H.gy + cy) * H.gx + cx], 1);
            }
        }
        __syncthreads();
        if (radius != 0.0f || pass >= 8) break;      // kernel argument: uniform
        int ne = 0;
        for (int c = tid; c < H.ncells; c += BQG_BUILD_T) ne += hist[c] > 0 ? 1 : 0;
        for (int off = 32; off >= 1; off >>= 1) ne += __shfl_xor(ne, off, 64);
        if (lane == 0 && ne) atomicAdd(&s_nonempty, ne);
        __syncthreads();
        if (tid == 0) {
            s_again = 0;
            if ((long long)s_nonempty * 2 < n) {
                const float h = s_h * 0.79370052598f;   // 2^(-1/3): half the cell volume
                int g[3];
                long long total = 1;
                for (int a = 0; a < 3; ++a) {
                    const float q = s_ext[a] / h;
                    g[a] = q < 1023.0f ? (int)q + 1 : 1024;
                    if (!(q == q)) g[a] = 1;
                    total *= g[a];
                }
                if (total <= BQG_CAP && h > 1e-30f) {
                    s_h = h;
                    sh.inv_h = 1.0f / h;
                    sh.gx = g[0]; sh.gy = g[1]; sh.gz = g[2];
                    sh.ncells = g[0] * g[1] * g[2];
                    s_again = 1;
                }
            }
        }
        __syncthreads();
        if (!s_again) break;
        H = sh;
        __syncthreads();
    }
    BQG_STAMP(1);   // histogram
    // points per OCCUPIED cell -> header float 8: the query kernel picks its form per cloud from it (a handful of candidates per
    // ball: one centre per lane; dense scenes: four centres per wave with the whole-wave path behind it)
    if (want_density) {   // (kernel argument: uniform; ~1 us)
        int ne = 0;
        for (int c = tid; c < H.ncells; c += BQG_BUILD_T) ne += hist[c] > 0 ? 1 : 0;
        for (int off = 32; off >= 1; off >>= 1) ne += __shfl_xor(ne, off, 64);
        if (tid == 0) s_nonempty = 0;
        __syncthreads();
        if (lane == 0 && ne) atomicAdd(&s_nonempty, ne);
        __syncthreads();
    }
    if (tid == 0) {
        float *hp = hdr_all + (size_t)b * BQG_HDR;
        hp[8] = want_density ? (float)n / (float)(s_nonempty > 0 ? s_nonempty : 1) : 1e30f;   // not measured: "dense" (the quad form)
        hp[0] = H.minx; hp[1] = H.miny; hp[2] = H.minz; hp[3] = H.inv_h;
        reinterpret_cast<int *>(hp)[4] = H.gx; reinterpret_cast<int *>(hp)[5] = H.gy;
        reinterpret_cast<int *>(hp)[6] = H.gz; reinterpret_cast<int *>(hp)[7] = H.ncells;
    }

    // ---- exclusive scan of hist[0..ncells) -> hist (running cursor) and cell_start (global)
    const int per = (H.ncells + BQG_BUILD_T - 1) / BQG_BUILD_T;
    const int c0 = tid * per, c1 = min(c0 + per, H.ncells);
    int local = 0;
    for (int c = c0; c < c1; ++c) local += hist[c];
    int incl = local;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int v = lane < BQG_BUILD_T / 64 ? wsum[lane] : 0;
        int inc = v;
        for (int off = 1; off < BQG_BUILD_T / 64; off <<= 1) {
            const int t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane < BQG_BUILD_T / 64) wsum[lane] = inc - v;  // exclusive wave offsets
    }
    __syncthreads();
    int run = wsum[wave] + incl - local;
    for (int c = c0; c < c1; ++c) {
        const int cnt = hist[c];
        hist[c] = run;
        run += cnt;
    }
    if (tid == 0) cell_start[H.ncells] = n;
    __syncthreads();
    // the starts leave through coalesced stores (a thread's own block of cells is `per` words apart from its neighbour's:
    // written from the loop above, the 14 k cell starts of a KITTI-range cloud took 4 of the scan's 6 us)
    for (int c = tid; c < H.ncells; c += BQG_BUILD_T) {
        const int v = hist[c];
        cell_start[c] = v;
        if constexpr (SPLIT) cursor_all[(size_t)b * (BQG_CAP + 1) + c] = v;
    }
    if constexpr (SPLIT) return;
    __syncthreads();
    BQG_STAMP(2);   // scan

    // ---- scatter into cell order (order inside a cell is arbitrary; the query does not depend on it)
    if constexpr (CACHED) {
#pragma unroll
        for (int i = 0; i < BQG_PPT; ++i) {
            const int k = tid + i * BQG_BUILD_T;
            if (k < n) {
                const int cx = cell_of(px[i], H.minx, H.inv_h, H.gx);
                const int cy = cell_of(py[i], H.miny, H.inv_h, H.gy);
                const int cz = cell_of(pz[i], H.minz, H.inv_h, H.gz);
                const int slot = atomicAdd(&hist[(cz * H.gy + cy) * H.gx + cx], 1);
                sorted[slot] = make_float4(px[i], py[i], pz[i], __int_as_float(k));
            }
        }
    } else {
        for (int k = tid; k < n; k += BQG_BUILD_T) {
            const float x = xyz[(size_t)k * 3 + 0], y = xyz[(size_t)k * 3 + 1], z = xyz[(size_t)k * 3 + 2];
            const int cx = cell_of(x, H.minx, H.inv_h, H.gx);
            const int cy = cell_of(y, H.miny, H.inv_h, H.gy);
            const int cz = cell_of(z, H.minz, H.inv_h, H.gz);
            const int slot = atomicAdd(&hist[(cz * H.gy + cy) * H.gx + cx], 1);
            sorted[slot] = make_float4(x, y, z, __int_as_float(k));
        }
    }
    BQG_STAMP(3);   // scatter issued (stores may still be in flight)
}

// Scatter half of the split build: (cloud, slab of points) per workgroup, a slot from the cloud's cursor table (device-scope
// atomic), the point written in cell order.  Order inside a cell is arbitrary, as in the one-kernel build: no query depends on it.
constexpr int BQG_SC_T = 256, BQG_SC_PPT = 4;
__global__ __launch_bounds__(BQG_SC_T) void bq_grid_scatter_kernel(int n, const float *__restrict__ xyz_all, const float *__restrict__ hdr_all,
                                                                   int *__restrict__ cursor_all, float4 *__restrict__ sorted_all) {
    const int b = blockIdx.y;
    const float *hp = hdr_all + (size_t)b * BQG_HDR;
    const float minx = hp[0], miny = hp[1], minz = hp[2], inv_h = hp[3];
    const int gx = reinterpret_cast<const int *>(hp)[4], gy = reinterpret_cast<const int *>(hp)[5], gz = reinterpret_cast<const int *>(hp)[6];
    const float *__restrict__ xyz = xyz_all + (size_t)b * n * 3;
    int *__restrict__ cursor = cursor_all + (size_t)b * (BQG_CAP + 1);
    float4 *__restrict__ sorted = sorted_all + (size_t)b * n;
    const int k0 = blockIdx.x * (BQG_SC_T * BQG_SC_PPT) + threadIdx.x;
    float px[BQG_SC_PPT], py[BQG_SC_PPT], pz[BQG_SC_PPT];
    int slot[BQG_SC_PPT];
#pragma unroll
    for (int i = 0; i < BQG_SC_PPT; ++i) {
        const int k = k0 + i * BQG_SC_T;
        px[i] = py[i] = pz[i] = 0.f;
        if (k < n) { px[i] = xyz[(size_t)k * 3 + 0]; py[i] = xyz[(size_t)k * 3 + 1]; pz[i] = xyz[(size_t)k * 3 + 2]; }
    }
#pragma unroll
    for (int i = 0; i < BQG_SC_PPT; ++i) {
        slot[i] = 0;
        if (k0 + i * BQG_SC_T < n) {
            const int cx = cell_of(px[i], minx, inv_h, gx), cy = cell_of(py[i], miny, inv_h, gy), cz = cell_of(pz[i], minz, inv_h, gz);
            slot[i] = atomicAdd(&cursor[(cz * gy + cy) * gx + cx], 1);
        }
    }
#pragma unroll
    for (int i = 0; i < BQG_SC_PPT; ++i) {
        const int k = k0 + i * BQG_SC_T;
        if (k < n) sorted[slot[i]] = make_float4(px[i], py[i], pz[i], __int_as_float(k));
    }
}

constexpr int BQG_QWAVES = 4;

struct BqGrid {
    float minx, miny, minz, inv_h;
    int gx, gy, gz;
    const int *__restrict__ cell_start;
    const float4 *__restrict__ sorted;
};

// One centre by a whole wave (any number of candidates): rows of the search box -> candidate numbering -> distance
// test -> hits ordered by index through registers (<= 64 candidates in <= 64 rows) or the wave's LDS bitmap.
__device__ __forceinline__ void bq_centre_wave(const BqGrid &G, float cx, float cy, float cz, float radius2, float absr,
                                               int nsample, int wpl, unsigned int *bm, int *out, int lane) {
    const float rx = search_halfwidth(cx, absr), ry = search_halfwidth(cy, absr), rz = search_halfwidth(cz, absr);
    const int x0 = cell_of(cx - rx, G.minx, G.inv_h, G.gx), x1 = cell_of(cx + rx, G.minx, G.inv_h, G.gx);
    const int y0 = cell_of(cy - ry, G.miny, G.inv_h, G.gy), y1 = cell_of(cy + ry, G.miny, G.inv_h, G.gy);
    const int z0 = cell_of(cz - rz, G.minz, G.inv_h, G.gz), z1 = cell_of(cz + rz, G.minz, G.inv_h, G.gz);
    // The (z, y) rows of the search box are x-contiguous runs of the sorted array.  All run bounds are
    // fetched at once (lane r = row r) and the candidates of all runs are numbered consecutively.
    const int ny = y1 - y0 + 1;
    const int nrows = ny * (z1 - z0 + 1);
    int hits = 0;          // wave-uniform
    bool used_bitmap = false;
    for (int r0 = 0; r0 < nrows; r0 += 64) {  // one pass unless the box spans more than 64 rows
        const int r = r0 + lane;
        int beg = 0, cntr = 0;
        if (r < nrows) {
            const int base = ((z0 + r / ny) * G.gy + (y0 + r % ny)) * G.gx;
            beg = G.cell_start[base + x0];
            cntr = G.cell_start[base + x1 + 1] - beg;
        }
        int incl = cntr;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        const int total = __shfl(incl, 63, 64);
        if (total == 0) continue;  // wave-uniform
        const int rows_here = min(64, nrows - r0);
        const int start = incl - cntr;   // exclusive prefix: first candidate number of this lane's row
        const bool single = nrows <= 64 && total <= 64;  // every candidate of this centre sits in one lane
        for (int t0 = 0; t0 < total; t0 += 64) {  // wave-uniform trip count
            const int t = t0 + lane;
            bool hit = false;
            int k = 0x7FFFFFFF;
            // candidate t lives in the LAST row whose first candidate number is <= t (empty rows share their
            // successor's number and lose to it): a uniform walk over the rows with scalar broadcasts, no LDS
            int rbeg = 0, rstart = 0;
            for (int r = 0; r < rows_here; ++r) {
                const int st = __builtin_amdgcn_readlane(start, r), bg = __builtin_amdgcn_readlane(beg, r);
                const bool ge = t >= st;
                rbeg = ge ? bg : rbeg;
                rstart = ge ? st : rstart;
            }
            if (t < total) {
                const float4 q = G.sorted[rbeg + (t - rstart)];
                const float d2 = sqdist(cx - q.x, cy - q.y, cz - q.z);
                hit = d2 < radius2;
                if (hit) k = __float_as_int(q.w);
            }
            const unsigned long long hm = __ballot(hit);
            if (single) {
                // fast path, registers only: a hit's output slot is the number of hits with a smaller index
                hits = __popcll(hm);
                if (hits > 0) {
                    int pos = 0, kmin = 0x7FFFFFFF;
                    for (unsigned long long mrest = hm; mrest != 0ull; mrest &= mrest - 1ull) {
                        const int other = __builtin_amdgcn_readlane(k, __ffsll((long long)mrest) - 1);
                        pos += other < k ? 1 : 0;
                        kmin = min(kmin, other);
                    }
                    if (hit && pos < nsample) out[pos] = k;
                    for (int l = hits + lane; l < nsample; l += 64) out[l] = kmin;  // ball_query_gpu.cu:41-45
                }
            } else {
                if (hit) atomicOr(&bm[k >> 5], 1u << (k & 31));
                hits += __popcll(hm);
                used_bitmap = true;
            }
        }
    }
    if (hits == 0) {   // empty ball: an all-zero row (what the reference's caller-side zero fill leaves, pointnet2_utils.py:218)
        for (int l = lane; l < nsample; l += 64) out[l] = 0;
        return;
    }
    if (!used_bitmap) return;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // LDS atomics of this wave are done before it reads them back

    // general path: lane owns bitmap words [lane*wpl, (lane+1)*wpl) = indices ascending with lane
    unsigned int *mine = bm + lane * wpl;
    int cnt = 0;
    for (int w = 0; w < wpl; ++w) cnt += __popc(mine[w]);
    int incl = cnt;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    int pos = incl - cnt;  // hits in lower lanes
    int first_local = -1;
    if (cnt > 0) {
        for (int w = 0; w < wpl; ++w) {
            unsigned int bits = mine[w];
            if (bits == 0u) continue;
            mine[w] = 0u;  // leave the bitmap clean for the next centre
            if (first_local < 0) first_local = (lane * wpl + w) * 32 + (__ffs(bits) - 1);
            while (bits != 0u && pos < nsample) {
                const int bit = __ffs(bits) - 1;
                bits &= bits - 1u;
                out[pos++] = (lane * wpl + w) * 32 + bit;
            }
        }
    }
    // slots beyond the hit count hold the first hit (ball_query_gpu.cu:41-45)
    const unsigned long long have = __ballot(cnt > 0);
    const int fl = __ffsll((long long)have) - 1;
    const int first = __shfl(first_local, fl, 64);
    for (int l = hits + lane; l < nsample; l += 64) out[l] = first;
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ BqGrid bq_grid_of(int b, int n, const float *__restrict__ hdr_all, const int *__restrict__ cell_start_all,
                                             const float4 *__restrict__ sorted_all) {
    const float *hp = hdr_all + (size_t)b * BQG_HDR;
    BqGrid G;
    G.minx = hp[0]; G.miny = hp[1]; G.minz = hp[2]; G.inv_h = hp[3];
    G.gx = reinterpret_cast<const int *>(hp)[4]; G.gy = reinterpret_cast<const int *>(hp)[5];
    G.gz = reinterpret_cast<const int *>(hp)[6];
    G.cell_start = cell_start_all + (size_t)b * (BQG_CAP + 1);
    G.sorted = sorted_all + (size_t)b * n;
    return G;
}

// One wave per centre (round 1's form; kept as the measured alternative: pdm_tune_bq_quad(0)).
__global__ __launch_bounds__(BQG_QWAVES * 64) void bq_grid_query_kernel(
    int n, int m, float radius, int nsample, int wpl, const float *__restrict__ new_xyz,
    const float *__restrict__ hdr_all, const int *__restrict__ cell_start_all,
    const float4 *__restrict__ sorted_all, int *__restrict__ idx) {
    extern __shared__ unsigned int bitmap_all[];  // BQG_QWAVES * wpl * 64 words
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int *bm = bitmap_all + (size_t)wave * wpl * 64;
    for (int w = lane; w < wpl * 64; w += 64) bm[w] = 0u;  // once; every centre leaves it clean
    const BqGrid G = bq_grid_of(b, n, hdr_all, cell_start_all, sorted_all);
    const float radius2 = radius * radius;  // ball_query_gpu.cu:29
    const float absr = fabsf(radius);
    // a wave walks centres j, j + stride, ...: the bitmap is cleared once per wave and left clean by every centre
    for (int j = blockIdx.x * BQG_QWAVES + wave; j < m; j += gridDim.x * BQG_QWAVES) {
        const float *c3 = new_xyz + ((size_t)b * m + j) * 3;
        bq_centre_wave(G, c3[0], c3[1], c3[2], radius2, absr, nsample, wpl, bm, idx + ((size_t)b * m + j) * nsample, lane);
    }
}

// Four centres per wave: a centre with at most 16 rows in its search box and at most 16 candidates in them (the
// common case once the cell follows the radius: a handful of candidates per ball) is answered by ONE 16-lane DPP
// row — run bounds, prefix sum, candidate -> run resolution, distance test and the by-index ranking of the hits
// are row-local (row_shr / row_newbcast), so one pass of instructions serves four centres; a centre that does not
// fit takes the whole-wave path above afterwards.  Same indices either way.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }

__device__ __forceinline__ void bq_quad_body(
    unsigned int *bitmap_all, int n, int m, float radius, int nsample, int wpl, const float *__restrict__ new_xyz,
    const float *__restrict__ hdr_all, const int *__restrict__ cell_start_all,
    const float4 *__restrict__ sorted_all, int *__restrict__ idx) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, q = lane >> 4;
    unsigned int *bm = bitmap_all + (size_t)wave * wpl * 64;
    for (int w = lane; w < wpl * 64; w += 64) bm[w] = 0u;
    const BqGrid G = bq_grid_of(b, n, hdr_all, cell_start_all, sorted_all);
    const float radius2 = radius * radius;
    const float absr = fabsf(radius);
    const int nquads = (m + 3) / 4;
    for (int jq = blockIdx.x * BQG_QWAVES + wave; jq < nquads; jq += gridDim.x * BQG_QWAVES) {
        const int j = jq * 4 + q;
        const bool live = j < m;
        const float *c3 = new_xyz + ((size_t)b * m + (live ? j : m - 1)) * 3;
        const float cx = c3[0], cy = c3[1], cz = c3[2];
        const float rx = search_halfwidth(cx, absr), ry = search_halfwidth(cy, absr), rz = search_halfwidth(cz, absr);
        const int x0 = cell_of(cx - rx, G.minx, G.inv_h, G.gx), x1 = cell_of(cx + rx, G.minx, G.inv_h, G.gx);
        const int y0 = cell_of(cy - ry, G.miny, G.inv_h, G.gy), y1 = cell_of(cy + ry, G.miny, G.inv_h, G.gy);
        const int z0 = cell_of(cz - rz, G.minz, G.inv_h, G.gz), z1 = cell_of(cz + rz, G.minz, G.inv_h, G.gz);
        const int ny = y1 - y0 + 1;
        const int nrows = ny * (z1 - z0 + 1);
        bool fast = live && nrows <= 16;
        int beg = 0, cntr = 0;
        if (fast && li < nrows) {
            const int base = ((z0 + li / ny) * G.gy + (y0 + li % ny)) * G.gx;
            beg = G.cell_start[base + x0];
            cntr = G.cell_start[base + x1 + 1] - beg;
        }
        // inclusive prefix sum inside the 16-lane row (zeros shift in)
        int incl = cntr;
        incl += dpp_i<0x111>(0, incl);   // row_shr:1
        incl += dpp_i<0x112>(0, incl);   // row_shr:2
        incl += dpp_i<0x114>(0, incl);   // row_shr:4
        incl += dpp_i<0x118>(0, incl);   // row_shr:8
        const int total = dpp_i<0x15F>(0, incl);   // row_newbcast:15
        fast = fast && total <= 16;
        const int start = incl - cntr;
        // candidate li of this centre lives in the last run whose first candidate number is <= li
        const int maxrows = __builtin_amdgcn_readfirstlane(
            max(max(__builtin_amdgcn_readlane(fast ? nrows : 0, 0), __builtin_amdgcn_readlane(fast ? nrows : 0, 16)),
                max(__builtin_amdgcn_readlane(fast ? nrows : 0, 32), __builtin_amdgcn_readlane(fast ? nrows : 0, 48))));
        int rbeg = 0, rstart = 0;
#define BQ_RUN(R)                                                                                  \
        if (R < maxrows) {                                                                         \
            const int st = dpp_i<0x150 + R>(0, start), bg = dpp_i<0x150 + R>(0, beg);              \
            const bool ge = li >= st && R < nrows;                                                 \
            rbeg = ge ? bg : rbeg;                                                                 \
            rstart = ge ? st : rstart;                                                             \
        }
        BQ_RUN(0) BQ_RUN(1) BQ_RUN(2) BQ_RUN(3) BQ_RUN(4) BQ_RUN(5) BQ_RUN(6) BQ_RUN(7)
        BQ_RUN(8) BQ_RUN(9) BQ_RUN(10) BQ_RUN(11) BQ_RUN(12) BQ_RUN(13) BQ_RUN(14) BQ_RUN(15)
#undef BQ_RUN
        int k = 0x7FFFFFFF;
        if (fast && li < total) {
            const float4 p4 = G.sorted[rbeg + (li - rstart)];
            const float d2 = sqdist(cx - p4.x, cy - p4.y, cz - p4.z);
            if (d2 < radius2) k = __float_as_int(p4.w);
        }
        // rank of a hit = number of hits of its centre with a smaller index; kmin = the first hit
        const int maxtot = __builtin_amdgcn_readfirstlane(
            max(max(__builtin_amdgcn_readlane(fast ? total : 0, 0), __builtin_amdgcn_readlane(fast ? total : 0, 16)),
                max(__builtin_amdgcn_readlane(fast ? total : 0, 32), __builtin_amdgcn_readlane(fast ? total : 0, 48))));
        int pos = 0, kmin = 0x7FFFFFFF, hits = 0;
#define BQ_RANK(R)                                                \
        if (R < maxtot) {                                         \
            const int other = dpp_i<0x150 + R>(0x7FFFFFFF, k);    \
            pos += other < k ? 1 : 0;                             \
            hits += other != 0x7FFFFFFF ? 1 : 0;                  \
            kmin = min(kmin, other);                              \
        }
        BQ_RANK(0) BQ_RANK(1) BQ_RANK(2) BQ_RANK(3) BQ_RANK(4) BQ_RANK(5) BQ_RANK(6) BQ_RANK(7)
        BQ_RANK(8) BQ_RANK(9) BQ_RANK(10) BQ_RANK(11) BQ_RANK(12) BQ_RANK(13) BQ_RANK(14) BQ_RANK(15)
#undef BQ_RANK
        if (fast) {
            int *out = idx + ((size_t)b * m + j) * nsample;
            if (k != 0x7FFFFFFF && pos < nsample) out[pos] = k;
            // ball_query_gpu.cu:41-45: pad with the first hit; an empty ball's row is all zeros (the reference's caller-side fill)
            for (int l = hits + li; l < nsample; l += 16) out[l] = hits > 0 ? kmin : 0;
        }
        // centres that did not fit a DPP row: whole-wave path, one after the other
        unsigned long long slow = __ballot(live && !fast) & 0x0001000100010001ull;
        while (slow != 0ull) {   // wave-uniform
            const int src = __ffsll((long long)slow) - 1;
            slow &= slow - 1ull;
            const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cx), src));
            const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cy), src));
            const float sz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cz), src));
            bq_centre_wave(G, sx, sy, sz, radius2, absr, nsample, wpl, bm, idx + ((size_t)b * m + jq * 4 + (src >> 4)) * nsample, lane);
        }
    }
}

__global__ __launch_bounds__(BQG_QWAVES * 64) void bq_grid_query4_kernel(
    int n, int m, float radius, int nsample, int wpl, const float *__restrict__ new_xyz,
    const float *__restrict__ hdr_all, const int *__restrict__ cell_start_all,
    const float4 *__restrict__ sorted_all, int *__restrict__ idx) {
    extern __shared__ unsigned int bitmap_all[];
    bq_quad_body(bitmap_all, n, m, radius, nsample, wpl, new_xyz, hdr_all, cell_start_all, sorted_all, idx);
}

// One centre per LANE (round 4).  The quad kernel above spends ~400 wave-instructions on four centres (run table,
// prefix sums, candidate -> run resolution and the by-index ranking as DPP chains) and is bound by instruction issue:
// 131072 centres of an SA1 call = 13 M wave-instructions = 30 us.  Here a lane owns a centre outright: the <= BQL_ROWS
// (z, y) rows of its search box are held as (first candidate number, array offset) pairs in registers, the candidates
// of all rows are numbered consecutively and fetched BQL_U at a time (independent 16-byte loads, all in flight), and
// the hits go into the lane's own LDS list kept in ascending index order (insertion; a full list only takes a smaller
// index) — so one pass of instructions serves 64 centres and the answer is still the first `nsample` hits BY INDEX,
// whatever the visiting order.  A centre with more than `heavy` candidates (dense scenes) is left to the whole-wave
// path (bq_centre_wave: 64 candidates per step, bitmap) afterwards.  Same indices on every path.
#ifndef BQL_U_N
#define BQL_U_N 4
#endif
constexpr int BQL_ROWS = 9, BQL_U = BQL_U_N;   // 9 = the 3 x 3 rows of a box that spans three cells per axis (an MSG level's larger radius on the
                                         // grid sized for its smaller one); more rows than that are walked in batches of 9
__device__ __forceinline__ void bq_lane_body(
    unsigned int *bitmap_all, int n, int m, float radius, int nsample, int wpl, int heavy, int cpw, const float *__restrict__ new_xyz,
    const float *__restrict__ hdr_all, const int *__restrict__ cell_start_all,
    const float4 *__restrict__ sorted_all, int *__restrict__ idx) {
    // cpw = centres per wave (16 / 32 / 64: lanes past it only take part in the whole-wave path).  Fewer centres per wave = more
    // waves for the same centres: the lane phase is a chain of dependent loads, and the heavy centres of a wave are answered one
    // after the other (dense lidar scenes: 310 us for an SA1 call at 64 per wave against 118 for the quad kernel).
    // bitmap_all: per wave ONE region of max(wpl * 64, cpw * LS) words that holds the cpw hit lists of LS words during the lane
    // phase of a group of centres and the whole-wave path's bitmap behind it (cleared when a group has heavy centres): with a
    // region each, the workgroup took twice the LDS and the quad form sharing the kernel lost residency (693 against 632 us on
    // lidar-like clouds)
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int LS = nsample | 1;                    // odd stride: slot s of the 64 lanes' lists spreads over the banks
    const int RW = max(wpl * 64, cpw * LS);
    unsigned int *bm = bitmap_all + (size_t)wave * RW;
    int *list = reinterpret_cast<int *>(bm) + (size_t)(lane < cpw ? lane : 0) * LS;
    const BqGrid G = bq_grid_of(b, n, hdr_all, cell_start_all, sorted_all);
    const float radius2 = radius * radius;
    const float absr = fabsf(radius);
    const int ngroups = (m + cpw - 1) / cpw;
    const bool vec_out = (nsample & 3) == 0 && (reinterpret_cast<uintptr_t>(idx) & 15) == 0;
    for (int jg = blockIdx.x * BQG_QWAVES + wave; jg < ngroups; jg += gridDim.x * BQG_QWAVES) {
        const int j = jg * cpw + lane;
        const bool live = lane < cpw && j < m;
        const float *c3 = new_xyz + ((size_t)b * m + (live ? j : m - 1)) * 3;
        const float cx = c3[0], cy = c3[1], cz = c3[2];
        const float rx = search_halfwidth(cx, absr), ry = search_halfwidth(cy, absr), rz = search_halfwidth(cz, absr);
        const int x0 = cell_of(cx - rx, G.minx, G.inv_h, G.gx), x1 = cell_of(cx + rx, G.minx, G.inv_h, G.gx);
        const int y0 = cell_of(cy - ry, G.miny, G.inv_h, G.gy), y1 = cell_of(cy + ry, G.miny, G.inv_h, G.gy);
        const int z0 = cell_of(cz - rz, G.minz, G.inv_h, G.gz), z1 = cell_of(cz + rz, G.minz, G.inv_h, G.gz);
        const int ny = y1 - y0 + 1;
        const int nrows = live ? ny * (z1 - z0 + 1) : 0;
        bool slow = false;
        int c = 0, seen = 0;                             // hits in the list (<= nsample), candidates visited so far
        for (int r0 = 0; __any(!slow && r0 < nrows); r0 += BQL_ROWS) {
            const bool on = !slow && r0 < nrows;
            // run bounds of the batch's rows at once: 2 * BQL_ROWS independent loads
            int rb[BQL_ROWS], re[BQL_ROWS];
#pragma unroll
            for (int r = 0; r < BQL_ROWS; ++r) {
                rb[r] = 0; re[r] = 0;
                if (on && r0 + r < nrows) {
                    const int rr = r0 + r;
                    const int base = ((z0 + rr / ny) * G.gy + (y0 + rr % ny)) * G.gx;
                    rb[r] = G.cell_start[base + x0];
                    re[r] = G.cell_start[base + x1 + 1];
                }
            }
            // candidate numbering: row r holds candidates [pre[r], pre[r + 1]); off[r] = array index of candidate t minus t
            int pre[BQL_ROWS], off[BQL_ROWS], total = 0;
#pragma unroll
            for (int r = 0; r < BQL_ROWS; ++r) {
                pre[r] = r0 + r < nrows ? total : 0x7FFFFFFF;     // rows past the box never match
                off[r] = rb[r] - total;
                total += re[r] - rb[r];
            }
            seen += total;
            if (on && seen > heavy) slow = true;              // too many candidates for a lane: the whole-wave path redoes this centre
            if (slow || !on) total = 0;
            for (int t0 = 0; __any(t0 < total); t0 += BQL_U) {
                float4 q[BQL_U];
#pragma unroll
                for (int u = 0; u < BQL_U; ++u) {
                    const int t = t0 + u;
                    int o = off[0];
#pragma unroll
                    for (int r = 1; r < BQL_ROWS; ++r) o = t >= pre[r] ? off[r] : o;
                    q[u] = t < total ? G.sorted[o + t] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < BQL_U; ++u) {
                    const float d2 = sqdist(cx - q[u].x, cy - q[u].y, cz - q[u].z);
                    if (t0 + u < total && d2 < radius2) {
                        const int k = __float_as_int(q[u].w);
                        int i = -1;
                        if (c < nsample) i = c++;
                        else if (k < list[nsample - 1]) i = nsample - 1;
                        if (i >= 0) {
                            while (i > 0) {
                                const int prev = list[i - 1];
                                if (prev < k) break;
                                list[i] = prev;
                                --i;
                            }
                            list[i] = k;
                        }
                    }
                }
            }
        }
        if (live && !slow) {
            // ball_query_gpu.cu:41-45: slots past the hits hold the first hit; an empty ball's row is all zeros
            int *out = idx + ((size_t)b * m + j) * nsample;
            const int first = c > 0 ? list[0] : 0;
            if (vec_out) {
                for (int s0 = 0; s0 < nsample; s0 += 4) {
                    int4 v;
                    v.x = s0 < c ? list[s0] : first; v.y = s0 + 1 < c ? list[s0 + 1] : first;
                    v.z = s0 + 2 < c ? list[s0 + 2] : first; v.w = s0 + 3 < c ? list[s0 + 3] : first;
                    *reinterpret_cast<int4 *>(out + s0) = v;
                }
            } else {
                for (int s0 = 0; s0 < nsample; ++s0) out[s0] = s0 < c ? list[s0] : first;
            }
        }
        // heavy centres: whole-wave path, one after the other
        unsigned long long todo = __ballot(slow);
        if (todo != 0ull) {      // the lists of this group are written out: the region becomes the (clean) bitmap
            __builtin_amdgcn_wave_barrier();
            for (int w = lane; w < wpl * 64; w += 64) bm[w] = 0u;
            __builtin_amdgcn_wave_barrier();
        }
        while (todo != 0ull) {   // wave-uniform
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cx), src));
            const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cy), src));
            const float sz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cz), src));
            bq_centre_wave(G, sx, sy, sz, radius2, absr, nsample, wpl, bm, idx + ((size_t)b * m + jg * cpw + src) * nsample, lane);
        }
    }
}

__global__ __launch_bounds__(BQG_QWAVES * 64) void bq_grid_query_lane_kernel(
    int n, int m, float radius, int nsample, int wpl, int heavy, int cpw, const float *__restrict__ new_xyz,
    const float *__restrict__ hdr_all, const int *__restrict__ cell_start_all,
    const float4 *__restrict__ sorted_all, int *__restrict__ idx) {
    extern __shared__ unsigned int bitmap_all[];
    bq_lane_body(bitmap_all, n, m, radius, nsample, wpl, heavy, cpw, new_xyz, hdr_all, cell_start_all, sorted_all, idx);
}

// The form is chosen per CLOUD from the density the build left in its header (points per occupied cell): sparse clouds (a handful
// of candidates per ball: BASELINE's uniform KITTI-range clouds) take the lane form, dense ones (lidar-like: hundreds of
// candidates in the near field) the quad form — measured on the API-exact block at bs = 32: uniform 473 (quad) / 452 (lane) us,
// lidar-like 642 (quad) / 713-943 (lane).  The grid is sized for the quad form; surplus workgroups of the lane form leave at once.
#ifndef BQL_MINWG
#define BQL_MINWG 7
#endif
#if BQL_MINWG
__global__ __launch_bounds__(BQG_QWAVES * 64, BQL_MINWG) void bq_grid_query_auto_kernel(
#else
__global__ __launch_bounds__(BQG_QWAVES * 64) void bq_grid_query_auto_kernel(
#endif
    int n, int m, float radius, int nsample, int wpl, int heavy, int cpw, float dense_ppc, const float *__restrict__ new_xyz,
    const float *__restrict__ hdr_all, const int *__restrict__ cell_start_all,
    const float4 *__restrict__ sorted_all, int *__restrict__ idx) {
    extern __shared__ unsigned int bitmap_all[];
    const float ppc = hdr_all[(size_t)blockIdx.y * BQG_HDR + 8];
    if (ppc >= dense_ppc) bq_quad_body(bitmap_all, n, m, radius, nsample, wpl, new_xyz, hdr_all, cell_start_all, sorted_all, idx);
    else bq_lane_body(bitmap_all, n, m, radius, nsample, wpl, heavy, cpw, new_xyz, hdr_all, cell_start_all, sorted_all, idx);
}

// Query form.  1 (default): four centres per wave (round 2's kernel).  2: one centre per lane (round 4).  3: per cloud by the density
// the build measures (lane for sparse, quad for dense clouds).  0: one wave per centre (round 1).  All return the same indices.
// Measured on the API-exact block of the bench (8 ball_query + 16 group_points, bs = 32, same process, profiles/r04i_ball_query_forms.txt):
// uniform KITTI-range clouds quad 462-473 us, lane 456-462, by density 461-477; lidar-like clouds quad 623-634, lane 716-960 (a
// wave answers its heavy centres one after the other), by density 683-704 (FPS-sampled levels of a dense scene count as sparse by
// points per cell and still hold dozens of candidates per ball).  2-3 % on the sparse case against 10-50 % on the dense one: the
// quad form stays the default.
static int g_bq_quad = 1;
// clouds of at least this many points (<= 16384) scatter from many workgroups; 0 = never (the default: MEASURED SLOWER — the
// cursor atomics of 64 lanes land in 64 different lines and run at ~20 G atomics/s chip-wide, so bq_grid_scatter_kernel takes
// 28 us for 32 x 16384 points where the one-workgroup scatter phase takes 8.5; profiles/r04d_api_block_split_build.txt)
static int g_grid_split_min_n = 0;
int launch_grid_build(hipStream_t stream, int b, int n, float radius, int max_cells, const float *xyz,
                      const GridWs &ws) {
    if (max_cells > BQG_CAP) max_cells = BQG_CAP;
    if (max_cells < 1) max_cells = 1;
    if (g_grid_split_min_n > 0 && n >= g_grid_split_min_n && n <= BQG_PPT * BQG_BUILD_T) {
        hipLaunchKernelGGL((bq_grid_build_kernel<true, true>), dim3(b), dim3(BQG_BUILD_T), 0, stream, n, radius, max_cells, xyz,
                           ws.hdr, ws.cell_start, ws.sorted, ws.cursor, g_bq_quad == 3);
        hipLaunchKernelGGL(bq_grid_scatter_kernel, dim3(divup(n, BQG_SC_T * BQG_SC_PPT), b), dim3(BQG_SC_T), 0, stream, n, xyz, ws.hdr,
                           ws.cursor, ws.sorted);
        return check_launch("grid_build(split)");
    }
    const int dens = g_bq_quad == 3;
    if (n <= BQG_PPT * BQG_BUILD_T)
        hipLaunchKernelGGL(bq_grid_build_kernel<true>, dim3(b), dim3(BQG_BUILD_T), 0, stream, n, radius, max_cells, xyz,
                           ws.hdr, ws.cell_start, ws.sorted, nullptr, dens);
    else
        hipLaunchKernelGGL(bq_grid_build_kernel<false>, dim3(b), dim3(BQG_BUILD_T), 0, stream, n, radius, max_cells, xyz,
                           ws.hdr, ws.cell_start, ws.sorted, nullptr, dens);
    return check_launch("grid_build");
}

}  // namespace pdm

using namespace pdm;

extern "C" size_t pdm_ball_query_grid_workspace_bytes(int b, int n) {
    return grid_workspace_bytes(b, n);
}

extern "C" int pdm_tune_bq_quad(int on) { const int old = pdm::g_bq_quad; pdm::g_bq_quad = on < 0 ? 0 : on > 3 ? 3 : on; return old; }
using pdm::g_bq_quad;
static float g_bq_dense_ppc = 3.0f;   // form 3: points per occupied cell from which a cloud counts as dense
extern "C" int pdm_tune_bq_dense_ppc(int hundredths) { const int old = (int)(g_bq_dense_ppc * 100.0f + 0.5f); if (hundredths >= 0) g_bq_dense_ppc = hundredths / 100.0f; return old; }
extern "C" int pdm_tune_grid_split(int min_n) { const int old = pdm::g_grid_split_min_n; pdm::g_grid_split_min_n = min_n; return old; }
static int g_bq_cpw = 16;    // lane form: centres per wave (16, 32 or 64)
extern "C" int pdm_tune_bq_cpw(int v) { const int old = g_bq_cpw; if (v == 16 || v == 32 || v == 64) g_bq_cpw = v; return old; }
static int g_bq_heavy = 96;  // lane form: a centre with more candidates than this goes to the whole-wave path
extern "C" int pdm_tune_bq_heavy(int v) { const int old = g_bq_heavy; if (v >= 0) g_bq_heavy = v; return old; }

// The search grid of a point set, built once and shared by every query over it: both radii of an SA level's ball
// queries and the three-NN of the FP module whose known set it is.  Any grid gives exact answers (the cell size only
// decides how many candidates a query visits); radius_hint sizes the cells for ball queries of about that radius,
// radius_hint == 0 asks for about two points per occupied cell (nearest-neighbour use).
extern "C" int pdm_grid_build(void *stream, int b, int n, float radius_hint, const float *xyz, void *workspace,
                              size_t workspace_bytes) {
    PDM_REQUIRE(b >= 0 && n >= 0, PDM_E_BADARG, "grid_build: negative size b=%d n=%d", b, n);
    if (b == 0 || n == 0) return 0;
    PDM_REQUIRE(xyz && workspace, PDM_E_BADARG, "grid_build: null pointer");
    PDM_REQUIRE(b <= 65535, PDM_E_TOOLARGE, "grid_build: b=%d exceeds grid", b);
    PDM_REQUIRE(workspace_bytes >= grid_workspace_bytes(b, n), PDM_E_BADARG, "grid_build: workspace of %zu bytes, need %zu",
                workspace_bytes, grid_workspace_bytes(b, n));
    const GridWs ws = grid_carve(workspace, b, n);
    return launch_grid_build(as_stream(stream), b, n, radius_hint, radius_hint == 0.0f ? (n / 2 > 8 ? n / 2 : 8) : BQG_CAP, xyz, ws);
}

static int bq_query_launch(void *stream, int b, int n, int m, float radius, int nsample, const float *new_xyz, int *idx,
                           const GridWs &ws) {
    const int wpl = (n + 2047) / 2048;  // bitmap words per lane: 64 lanes x wpl words x 32 bits >= n
    const size_t lds = (size_t)BQG_QWAVES * wpl * 64 * sizeof(unsigned int);
    PDM_REQUIRE(lds <= 64 * 1024, PDM_E_TOOLARGE, "ball_query_grid: n=%d needs %zu bytes of LDS bitmap", n, lds);
    const int cpw = g_bq_cpw;
    const size_t lists = (size_t)BQG_QWAVES * cpw * (nsample | 1) * sizeof(int);
    const size_t lds_lane = lds > lists ? lds : lists;     // one region per wave: hit lists, then the whole-wave path's bitmap
    if (g_bq_quad == 3 && lds_lane <= 64 * 1024) {
        const int per_sample = divup(divup(m, 4), BQG_QWAVES);    // the quad form's work items; the lane form needs a quarter (cpw = 16) or less
        const int cap = (256 * 16 + b - 1) / b;
        dim3 grid(per_sample < cap ? per_sample : cap, b);
        hipLaunchKernelGGL(bq_grid_query_auto_kernel, grid, dim3(BQG_QWAVES * 64), lds_lane, as_stream(stream), n, m, radius, nsample, wpl,
                           g_bq_heavy, cpw, g_bq_dense_ppc, new_xyz, ws.hdr, ws.cell_start, ws.sorted, idx);
        return check_launch("ball_query_grid(query, auto)");
    }
    if (g_bq_quad == 2 && lds_lane <= 64 * 1024) {
        const int per_sample = divup(divup(m, cpw), BQG_QWAVES);
        const int cap = (256 * 16 + b - 1) / b;
        dim3 grid(per_sample < cap ? per_sample : cap, b);
        hipLaunchKernelGGL(bq_grid_query_lane_kernel, grid, dim3(BQG_QWAVES * 64), lds_lane, as_stream(stream), n, m, radius, nsample, wpl,
                           g_bq_heavy, cpw, new_xyz, ws.hdr, ws.cell_start, ws.sorted, idx);
        return check_launch("ball_query_grid(query, lane)");
    }
    const int units = g_bq_quad ? divup(m, 4) : m;           // work items per sample: quads of centres or centres
    const int per_sample = divup(units, BQG_QWAVES);
    const int cap = (256 * 16 + b - 1) / b;  // about 16 four-wave workgroups per CU in total; waves loop beyond that
    dim3 grid(per_sample < cap ? per_sample : cap, b);
    if (g_bq_quad)     // (1, or 2 / 3 where the lane form's lists do not fit the LDS)
        hipLaunchKernelGGL(bq_grid_query4_kernel, grid, dim3(BQG_QWAVES * 64), lds, as_stream(stream), n, m, radius, nsample, wpl,
                           new_xyz, ws.hdr, ws.cell_start, ws.sorted, idx);
    else
        hipLaunchKernelGGL(bq_grid_query_kernel, grid, dim3(BQG_QWAVES * 64), lds, as_stream(stream), n, m, radius, nsample, wpl,
                           new_xyz, ws.hdr, ws.cell_start, ws.sorted, idx);
    return check_launch("ball_query_grid(query)");
}

// Ball query against a grid pdm_grid_build left in `workspace` (same b, n): identical indices to pdm_ball_query.
extern "C" int pdm_ball_query_grid_prebuilt(void *stream, int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                            int *idx, const void *workspace, size_t workspace_bytes) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, PDM_E_BADARG,
                "ball_query_grid: negative size b=%d n=%d m=%d nsample=%d", b, n, m, nsample);
    if (b == 0 || m == 0 || nsample == 0 || n == 0) return 0;
    PDM_REQUIRE(new_xyz && idx && workspace, PDM_E_BADARG, "ball_query_grid: null pointer");
    PDM_REQUIRE(b <= 65535, PDM_E_TOOLARGE, "ball_query_grid: b=%d exceeds grid", b);
    PDM_REQUIRE(workspace_bytes >= grid_workspace_bytes(b, n), PDM_E_BADARG, "ball_query_grid: workspace of %zu bytes, need %zu",
                workspace_bytes, grid_workspace_bytes(b, n));
    return bq_query_launch(stream, b, n, m, radius, nsample, new_xyz, idx, grid_carve(const_cast<void *>(workspace), b, n));
}

extern "C" int pdm_ball_query_grid(void *stream, int b, int n, int m, float radius, int nsample,
                                   const float *new_xyz, const float *xyz, int *idx, void *workspace,
                                   size_t workspace_bytes) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, PDM_E_BADARG,
                "ball_query_grid: negative size b=%d n=%d m=%d nsample=%d", b, n, m, nsample);
    if (b == 0 || m == 0 || nsample == 0 || n == 0) return 0;
    PDM_REQUIRE(new_xyz && xyz && idx && workspace, PDM_E_BADARG, "ball_query_grid: null pointer");
    PDM_REQUIRE(b <= 65535, PDM_E_TOOLARGE, "ball_query_grid: b=%d exceeds grid", b);
    PDM_REQUIRE(workspace_bytes >= pdm_ball_query_grid_workspace_bytes(b, n), PDM_E_BADARG,
                "ball_query_grid: workspace of %zu bytes, need %zu", workspace_bytes,
                pdm_ball_query_grid_workspace_bytes(b, n));
    const GridWs ws = grid_carve(workspace, b, n);
    int rc = launch_grid_build(as_stream(stream), b, n, radius, BQG_CAP, xyz, ws);
    if (rc) return rc;
    return bq_query_launch(stream, b, n, m, radius, nsample, new_xyz, idx, ws);
}
