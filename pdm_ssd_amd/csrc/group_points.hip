// group_points (forward / backward) and the fused QueryAndGroup gather for gfx950.
//
// Semantics: /root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/group_points_gpu.cu
// (forward :53-72, backward :14-31) and pointnet2_utils.py:241-264 for the fused form.
//
// The op is a pure HBM-write stream: out (B,C,M*ns) is written once, the source rows (N floats
// per (b,c)) are small enough to stay in L2.  Design: a thread owns FOUR consecutive outputs of
// the flattened (M*ns) axis — one 16-byte index load, then for each channel of its channel group
// four L2-resident gathers and one 16-byte coalesced store.  The reference issues one 4-byte
// store per thread and re-reads the index for every channel.
#include "common.h"

namespace pdm {

constexpr int GP_THREADS = 256;
constexpr int GP_CG = 8;  // channels per workgroup (index registers reused across them)

// vectorised: L % 4 == 0, idx/out 16-byte aligned
// outputs are written once and read by another kernel much later: non-temporal stores keep them from displacing the index and
// source lines the gathers hit in L2 (pdm_tune_group_nt; measured with tools/diag/api_block_ab.py)
typedef float gp_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gp_store4(float *p, const float4 &v, int nt) {
    if (nt) __builtin_nontemporal_store(gp_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<gp_f4 *>(p));
    else *reinterpret_cast<float4 *>(p) = v;
}

__global__ __launch_bounds__(GP_THREADS) void group_points_v4_kernel(
    int c, int n, long long L, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out, int nt = 0) {
    const int b = blockIdx.z;
    const long long q = (long long)blockIdx.x * GP_THREADS + threadIdx.x;  // quad index
    if (q * 4 >= L) return;
    const int4 id = *reinterpret_cast<const int4 *>(idx + (size_t)b * L + q * 4);
    const int c0 = blockIdx.y * GP_CG;
    const int c1 = min(c0 + GP_CG, c);
    for (int ci = c0; ci < c1; ++ci) {
        const float *__restrict__ row = points + ((size_t)b * c + ci) * n;
        float4 v;
        v.x = row[id.x]; v.y = row[id.y]; v.z = row[id.z]; v.w = row[id.w];
        gp_store4(out + ((size_t)b * c + ci) * L + q * 4, v, nt);
    }
}

__global__ __launch_bounds__(GP_THREADS) void group_points_scalar_kernel(
    int c, int n, long long L, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out) {
    const int b = blockIdx.z;
    const long long l = (long long)blockIdx.x * GP_THREADS + threadIdx.x;
    if (l >= L) return;
    const int id = idx[(size_t)b * L + l];
    const int c0 = blockIdx.y * GP_CG;
    const int c1 = min(c0 + GP_CG, c);
    for (int ci = c0; ci < c1; ++ci)
        out[((size_t)b * c + ci) * L + l] = points[((size_t)b * c + ci) * n + id];
}

__global__ __launch_bounds__(GP_THREADS) void group_points_grad_kernel(
    int c, int n, long long L, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const long long l = (long long)blockIdx.x * GP_THREADS + threadIdx.x;
    if (l >= L) return;
    const int id = idx[(size_t)b * L + l];
    const int c0 = blockIdx.y * GP_CG;
    const int c1 = min(c0 + GP_CG, c);
    for (int ci = c0; ci < c1; ++ci)
        atomicAdd(grad_points + ((size_t)b * c + ci) * n + id, grad_out[((size_t)b * c + ci) * L + l]);
}

// Fused QueryAndGroup gather: out (B, 3+C, M, ns); blockIdx.y == 0 writes the three centred xyz
// channels, blockIdx.y >= 1 a group of feature channels.  Requires ns % 4 == 0 (a quad never
// straddles two centres) and 16-byte aligned idx/out.
__global__ __launch_bounds__(GP_THREADS) void query_group_v4_kernel(
    int c, int n, int m, int ns, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ features, const int *__restrict__ idx, float *__restrict__ out) {
    const int b = blockIdx.z;
    const long long L = (long long)m * ns;
    const long long q = (long long)blockIdx.x * GP_THREADS + threadIdx.x;
    if (q * 4 >= L) return;
    const int4 id = *reinterpret_cast<const int4 *>(idx + (size_t)b * L + q * 4);
    float *__restrict__ ob = out + (size_t)b * (3 + c) * L + q * 4;
    if (blockIdx.y == 0) {
        const int j = (int)((q * 4) / ns);
        const float *__restrict__ p = xyz + (size_t)b * n * 3;
        const float *__restrict__ ctr = new_xyz + ((size_t)b * m + j) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float ca = ctr[a];
            float4 v;
            v.x = p[(size_t)id.x * 3 + a] - ca;
            v.y = p[(size_t)id.y * 3 + a] - ca;
            v.z = p[(size_t)id.z * 3 + a] - ca;
            v.w = p[(size_t)id.w * 3 + a] - ca;
            *reinterpret_cast<float4 *>(ob + (size_t)a * L) = v;
        }
    } else {
        const int c0 = (blockIdx.y - 1) * GP_CG;
        const int c1 = min(c0 + GP_CG, c);
        for (int ci = c0; ci < c1; ++ci) {
            const float *__restrict__ row = features + ((size_t)b * c + ci) * n;
            float4 v;
            v.x = row[id.x]; v.y = row[id.y]; v.z = row[id.z]; v.w = row[id.w];
            *reinterpret_cast<float4 *>(ob + (size_t)(3 + ci) * L) = v;
        }
    }
}

__global__ __launch_bounds__(GP_THREADS) void query_group_scalar_kernel(
    int c, int n, int m, int ns, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ features, const int *__restrict__ idx, float *__restrict__ out) {
    const int b = blockIdx.z;
    const long long L = (long long)m * ns;
    const long long l = (long long)blockIdx.x * GP_THREADS + threadIdx.x;
    if (l >= L) return;
    const int id = idx[(size_t)b * L + l];
    float *__restrict__ ob = out + (size_t)b * (3 + c) * L + l;
    if (blockIdx.y == 0) {
        const int j = (int)(l / ns);
        for (int a = 0; a < 3; ++a)
            ob[(size_t)a * L] = xyz[((size_t)b * n + id) * 3 + a] - new_xyz[((size_t)b * m + j) * 3 + a];
    } else {
        const int c0 = (blockIdx.y - 1) * GP_CG;
        const int c1 = min(c0 + GP_CG, c);
        for (int ci = c0; ci < c1; ++ci)
            ob[(size_t)(3 + ci) * L] = features[((size_t)b * c + ci) * n + id];
    }
}

// LDS-staged forms.  With real (non-repeating) ball-query indices the direct kernel above is bound by the
// vector L1's line rate: a wave gather of 64 random 4-byte elements touches up to 64 cache lines (measured
// 1.9 TB/s at C=96, N=4096 with random indices vs 4.8 TB/s with the padded indices of sparse clouds).  Here a
// workgroup first copies its source rows into LDS with coalesced 16-byte loads (a (b,c) row is N floats) and
// gathers from there: LDS serves 64 random dwords in a few cycles.  The backward pass accumulates each row in
// LDS (ds_add_f32) and writes it once: no global atomics.
constexpr int GPL_THREADS = 256;

__global__ __launch_bounds__(GPL_THREADS) void group_points_lds_kernel(
    int c, int n, long long L, int rows_per_wg, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out, int nt = 0) {
    extern __shared__ float rows[];  // rows_per_wg x n
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * rows_per_wg;
    const int nr = min(rows_per_wg, c - c0);
    const float *__restrict__ src = points + ((size_t)b * c + c0) * n;
    const long long tot = (long long)nr * n;
    if ((tot & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        for (long long i = (long long)threadIdx.x * 4; i < tot; i += GPL_THREADS * 4)
            *reinterpret_cast<float4 *>(rows + i) = *reinterpret_cast<const float4 *>(src + i);
    } else {
        for (long long i = threadIdx.x; i < tot; i += GPL_THREADS) rows[i] = src[i];
    }
    __syncthreads();
    const int *__restrict__ ib = idx + (size_t)b * L;
    float *__restrict__ ob = out + ((size_t)b * c + c0) * L;
    if ((L & 3) == 0 && (reinterpret_cast<uintptr_t>(ib) & 15) == 0 && (reinterpret_cast<uintptr_t>(ob) & 15) == 0) {
        for (long long q = threadIdx.x; q < L / 4; q += GPL_THREADS) {
            const int4 id = *reinterpret_cast<const int4 *>(ib + q * 4);
            for (int r = 0; r < nr; ++r) {
                const float *row = rows + (size_t)r * n;
                float4 v;
                v.x = row[id.x]; v.y = row[id.y]; v.z = row[id.z]; v.w = row[id.w];
                gp_store4(ob + (size_t)r * L + q * 4, v, nt);
            }
        }
    } else {
        for (long long l = threadIdx.x; l < L; l += GPL_THREADS) {
            const int id = ib[l];
            for (int r = 0; r < nr; ++r) ob[(size_t)r * L + l] = rows[(size_t)r * n + id];
        }
    }
}

__global__ __launch_bounds__(GPL_THREADS) void group_points_grad_lds_kernel(
    int c, int n, long long L, int rows_per_wg, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points) {
    extern __shared__ float rows[];  // rows_per_wg x n accumulators
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * rows_per_wg;
    const int nr = min(rows_per_wg, c - c0);
    const long long tot = (long long)nr * n;
    for (long long i = threadIdx.x; i < tot; i += GPL_THREADS) rows[i] = 0.0f;
    __syncthreads();
    const int *__restrict__ ib = idx + (size_t)b * L;
    const float *__restrict__ gb = grad_out + ((size_t)b * c + c0) * L;
    for (long long l = threadIdx.x; l < L; l += GPL_THREADS) {
        const int id = ib[l];
        for (int r = 0; r < nr; ++r) atomicAdd(&rows[(size_t)r * n + id], gb[(size_t)r * L + l]);
    }
    __syncthreads();
    float *__restrict__ dst = grad_points + ((size_t)b * c + c0) * n;
    for (long long i = threadIdx.x; i < tot; i += GPL_THREADS) dst[i] += rows[i];  // caller-zeroed, accumulated
}

// Round 3 form of the LDS-staged gather.  What round 2's kernel lost (4.2-4.5 TB/s on every large launch) was not
// bandwidth per workgroup but GRANULARITY: one workgroup = (sample, row group) streaming the whole L axis, 1536 such
// units over 1280 resident slots is one full round plus a round that fills a fifth of the chip.  Here
//   * the grid is one-dimensional and a unit is (sample, row group, part of L): `lsplit` parts re-stage the same rows
//     (from L2: the sibling parts run on the same XCD) and make the units short enough for the last round to matter less;
//   * blockIdx -> unit is XCD-aware (workgroups w and w + 8 share an XCD's L2): XCD x works through samples x, x + 8, ...
//     so a sample's index list and rows are fetched into ONE L2;
//   * a lane holds UQ index quads per pass: UQ x 16 bytes of index loads in flight, then per row UQ x 4 LDS reads and
//     UQ 16-byte stores (a wave instruction stores 1 KB contiguous);
//   * the first pass's indices are requested BEFORE the rows are staged, so the two latencies overlap.
// Same values as every other form (pure data movement).
template <int T, int GPR_UQ, bool XCD>
__global__ __launch_bounds__(T) void group_points_rows_kernel(int c, int n, long long L, int rpw, int nrg, int lsplit, int B,
                                                              const float *__restrict__ points, const int *__restrict__ idx,
                                                              float *__restrict__ out, int nt = 0) {
    extern __shared__ float rows[];  // rpw x n
    const int per_b = nrg * lsplit;
    int b, rem;
    if (XCD && (B & 7) == 0) {
        const int x = blockIdx.x & 7, t = blockIdx.x >> 3;
        b = (t / per_b) * 8 + x;
        rem = t % per_b;
    } else {
        b = blockIdx.x / per_b;
        rem = blockIdx.x % per_b;
    }
    const int rg = rem / lsplit, lp = rem % lsplit;
    const int c0 = rg * rpw;
    const int nr = min(rpw, c - c0);
    const long long nq = L >> 2;
    const long long qper = (nq + lsplit - 1) / lsplit;
    const long long q_begin = (long long)lp * qper;
    const long long q_end = min(nq, q_begin + qper);
    const int4 *__restrict__ ib = reinterpret_cast<const int4 *>(idx + (size_t)b * L);
    int4 id[GPR_UQ];
    long long q0 = q_begin + threadIdx.x;
#pragma unroll
    for (int u = 0; u < GPR_UQ; ++u) {
        const long long q = q0 + (long long)u * T;
        id[u] = q < q_end ? ib[q] : make_int4(0, 0, 0, 0);
    }
    const float *__restrict__ src = points + ((size_t)b * c + c0) * n;
    const long long tot = (long long)nr * n;
    if ((tot & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        for (long long i = (long long)threadIdx.x * 4; i < tot; i += T * 4)
            *reinterpret_cast<float4 *>(rows + i) = *reinterpret_cast<const float4 *>(src + i);
    } else {
        for (long long i = threadIdx.x; i < tot; i += T) rows[i] = src[i];
    }
    __syncthreads();
    float4 *__restrict__ ob = reinterpret_cast<float4 *>(out + ((size_t)b * c + c0) * L);
    while (q0 < q_end) {
        for (int r = 0; r < nr; ++r) {
            const float *row = rows + (size_t)r * n;
#pragma unroll
            for (int u = 0; u < GPR_UQ; ++u) {
                const long long q = q0 + (long long)u * T;
                if (q < q_end) {
                    float4 v;
                    v.x = row[id[u].x]; v.y = row[id[u].y]; v.z = row[id[u].z]; v.w = row[id[u].w];
                    gp_store4(reinterpret_cast<float *>(ob + (size_t)r * nq + q), v, nt);
                }
            }
        }
        q0 += (long long)GPR_UQ * T;
#pragma unroll
        for (int u = 0; u < GPR_UQ; ++u) {
            const long long q = q0 + (long long)u * T;
            if (q < q_end) id[u] = ib[q];
        }
    }
}

// tuning knob (tools/diag/group_sweep.py): 0 = heuristics; else variant (1 rows kernel, 2 round-2 kernel, 3 rows kernel without
// the XCD unit order) | rpw << 4 | lsplit << 8 | (threads / 256) << 16 | quads per lane and pass << 20
static int g_gp_tune = 0;
static int g_gp_nt = 1;      // 1 (default): non-temporal stores of the grouped outputs (455 against 458-463 us on the target block)
extern "C" int pdm_tune_group_nt(int on) { const int old = g_gp_nt; g_gp_nt = on != 0; return old; }
// LDS-staged gathers request at least this many bytes of LDS per workgroup (0 = what the rows need).  A streaming gather fills
// every wave slot of the chip and is bound by HBM long before that: beside it a latency-bound kernel of another stream (a ball
// query, a grid build) waits for whole rounds of its workgroups.  64 KB = two workgroups per CU and 32 KB of LDS left over.
static int g_gp_lds_floor = 0;
extern "C" int pdm_tune_group_lds_floor(int bytes) { const int old = g_gp_lds_floor; if (bytes >= 0 && bytes <= 64 * 1024) g_gp_lds_floor = bytes; return old; }

// rows of n floats per workgroup: up to `budget` bytes of LDS (several workgroups per CU overlap one's staging
// with another's streaming), at most 8 (index registers reused across them); 0 = a row does not fit 64 KB
static inline int lds_rows_per_wg(int n, int c, size_t budget) {
    if (n <= 0 || (size_t)n * sizeof(float) > 64 * 1024) return 0;
    int r = (int)(budget / ((size_t)n * sizeof(float)));
    r = r < 1 ? 1 : r > 8 ? 8 : r;
    return r > c ? c : r;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace pdm

using namespace pdm;

extern "C" int pdm_tune_group_rows(int packed) { const int old = g_gp_tune; g_gp_tune = packed; return old; }

extern "C" int pdm_group_points(void *stream, int b, int c, int n, int npoints, int nsample,
                                const float *points, const int *idx, float *out) {
    PDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, PDM_E_BADARG,
                "group_points: negative size");
    const long long L = (long long)npoints * nsample;
    if (b == 0 || c == 0 || L == 0) return 0;
    PDM_REQUIRE(points && idx && out, PDM_E_BADARG, "group_points: null pointer");
    PDM_REQUIRE(b <= 65535 && divup(c, GP_CG) <= 65535, PDM_E_TOOLARGE, "group_points: b=%d c=%d exceed grid", b, c);
    int rpw = lds_rows_per_wg(n, c, 32 * 1024);
    // LDS staging pays when a row is re-used by many outputs and there are enough (b, c) rows to fill the chip.
    // Which form / decomposition: profiles/r03_group_points_sweep.txt (tools/diag/group_sweep.py, bs = 32).  The rows
    // kernel wins where a unit of round 2's kernel was long (rows of 16 KB: N = 4096) or very short (N = 256); for 4 KB
    // rows with L = 8192 round 2's kernel stays ahead (6.2 TB/s) and is kept.
    int variant = 1, lsplit = 1, threads = 256, uq = 4;
    const size_t row_bytes = (size_t)n * sizeof(float);
    if (row_bytes > 8 * 1024) {            // long rows: few rows fit, units are long -> big workgroups, L in two parts
        if (L >= 32768) { threads = 1024; rpw = (int)(64 * 1024 / row_bytes); lsplit = 2; }
        else { threads = 512; rpw = (int)(48 * 1024 / row_bytes); }
    } else if (row_bytes > 2 * 1024) {
        if (L >= 8192) variant = 2;
        else threads = 512;
    }
    rpw = rpw < 1 ? 1 : rpw > 8 ? 8 : rpw > c ? c : rpw;
    if (g_gp_tune) {
        variant = g_gp_tune & 15;
        const int r = (g_gp_tune >> 4) & 15, ls = (g_gp_tune >> 8) & 255, th = (g_gp_tune >> 16) & 7, u = (g_gp_tune >> 20) & 7;
        rpw = lds_rows_per_wg(n, c, 32 * 1024);
        if (r > 0 && (size_t)r * row_bytes <= 64 * 1024) rpw = r > c ? c : r;
        lsplit = ls > 0 ? ls : 1;
        threads = (th == 2 || th == 4) ? 256 * th : 256;
        uq = (u == 1 || u == 2) ? u : 4;
    }
    if ((variant == 1 || variant == 3) && rpw > 0 && L >= 4 * n && L % 4 == 0 && aligned16(idx) && aligned16(out) && aligned16(points)
        && (long long)b * divup(c, rpw) >= 512) {
        const int nrg = divup(c, rpw);
        const long long nq = L / 4;
        if ((long long)lsplit > nq) lsplit = (int)nq;
        const long long wgs = (long long)b * nrg * lsplit;
        PDM_REQUIRE(wgs <= 0x7fffffffll, PDM_E_TOOLARGE, "group_points: %lld workgroups", wgs);
        const size_t lds = std::max((size_t)rpw * row_bytes, (size_t)g_gp_lds_floor);
#define GPR_LAUNCH(T, U, X) hipLaunchKernelGGL((group_points_rows_kernel<T, U, X>), dim3((unsigned)wgs), dim3(T), lds, as_stream(stream), c, n, L, rpw, nrg, lsplit, b, points, idx, out, g_gp_nt)
        if (variant == 3) {            // diagnostic: plain unit order (no XCD grouping)
            if (threads == 256) GPR_LAUNCH(256, 4, false); else if (threads == 512) GPR_LAUNCH(512, 4, false); else GPR_LAUNCH(1024, 4, false);
        } else if (uq == 1) {
            if (threads == 256) GPR_LAUNCH(256, 1, true); else if (threads == 512) GPR_LAUNCH(512, 1, true); else GPR_LAUNCH(1024, 1, true);
        } else if (uq == 2) {
            if (threads == 256) GPR_LAUNCH(256, 2, true); else if (threads == 512) GPR_LAUNCH(512, 2, true); else GPR_LAUNCH(1024, 2, true);
        } else {
            if (threads == 256) GPR_LAUNCH(256, 4, true); else if (threads == 512) GPR_LAUNCH(512, 4, true); else GPR_LAUNCH(1024, 4, true);
        }
#undef GPR_LAUNCH
        return check_launch("group_points");
    }
    rpw = lds_rows_per_wg(n, c, 32 * 1024);   // round 2's kernel and its decomposition
    if (rpw > 0 && L >= 4 * n && (long long)b * divup(c, rpw) >= 512) {
        dim3 grid(divup(c, rpw), b);
        hipLaunchKernelGGL(group_points_lds_kernel, grid, dim3(GPL_THREADS), std::max((size_t)rpw * n * sizeof(float), (size_t)g_gp_lds_floor),
                           as_stream(stream), c, n, L, rpw, points, idx, out, g_gp_nt);
        return check_launch("group_points");
    }
    if (L % 4 == 0 && aligned16(idx) && aligned16(out)) {
        dim3 grid(divup(L / 4, GP_THREADS), divup(c, GP_CG), b);
        hipLaunchKernelGGL(group_points_v4_kernel, grid, dim3(GP_THREADS), 0, as_stream(stream), c, n,
                           L, points, idx, out, g_gp_nt);
    } else {
        dim3 grid(divup(L, GP_THREADS), divup(c, GP_CG), b);
        hipLaunchKernelGGL(group_points_scalar_kernel, grid, dim3(GP_THREADS), 0, as_stream(stream),
                           c, n, L, points, idx, out);
    }
    return check_launch("group_points");
}

extern "C" int pdm_group_points_grad(void *stream, int b, int c, int n, int npoints, int nsample,
                                     const float *grad_out, const int *idx, float *grad_points) {
    PDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, PDM_E_BADARG,
                "group_points_grad: negative size");
    const long long L = (long long)npoints * nsample;
    if (b == 0 || c == 0 || L == 0) return 0;
    PDM_REQUIRE(grad_out && idx && grad_points, PDM_E_BADARG, "group_points_grad: null pointer");
    PDM_REQUIRE(b <= 65535 && divup(c, GP_CG) <= 65535, PDM_E_TOOLARGE, "group_points_grad: exceeds grid");
    const int rpw = lds_rows_per_wg(n, c, 32 * 1024);
    if (rpw > 0 && L >= n && (long long)b * divup(c, rpw) >= 256) {
        dim3 grid_l(divup(c, rpw), b);
        hipLaunchKernelGGL(group_points_grad_lds_kernel, grid_l, dim3(GPL_THREADS), (size_t)rpw * n * sizeof(float),
                           as_stream(stream), c, n, L, rpw, grad_out, idx, grad_points);
        return check_launch("group_points_grad");
    }
    dim3 grid(divup(L, GP_THREADS), divup(c, GP_CG), b);
    hipLaunchKernelGGL(group_points_grad_kernel, grid, dim3(GP_THREADS), 0, as_stream(stream), c, n, L,
                       grad_out, idx, grad_points);
    return check_launch("group_points_grad");
}

extern "C" size_t pdm_group_points_grad_ws_bytes(int b, int npoints, int nsample, int n) {
    return csr_workspace_bytes(b, (long long)npoints * nsample, n);
}

// pdm_group_points_grad with a caller-provided workspace (pdm_group_points_grad_ws_bytes bytes): the scatter onto the n
// source points is inverted into per-cloud CSR lists, then accumulated without atomics (interpolate.hip).  Applies when a
// grad_out row (npoints * nsample floats) fits 128 KB of LDS and n <= 16384; otherwise the plain entry point is used.
extern "C" int pdm_group_points_grad_ws(void *stream, int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                        const int *idx, float *grad_points, void *workspace, size_t workspace_bytes) {
    PDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, PDM_E_BADARG, "group_points_grad_ws: negative size");
    const long long L = (long long)npoints * nsample;
    if (b == 0 || c == 0 || L == 0) return 0;
    PDM_REQUIRE(grad_out && idx && grad_points, PDM_E_BADARG, "group_points_grad_ws: null pointer");
    if (L > 32768 || !csr_form_applies(b, (int)L, L, n))
        return pdm_group_points_grad(stream, b, c, n, npoints, nsample, grad_out, idx, grad_points);
    PDM_REQUIRE(workspace && workspace_bytes >= pdm_group_points_grad_ws_bytes(b, npoints, nsample, n), PDM_E_BADARG,
                "group_points_grad_ws: workspace of %zu bytes, need %zu", workspace_bytes,
                pdm_group_points_grad_ws_bytes(b, npoints, nsample, n));
    return csr_scatter_grad_launch(stream, "group_points_grad_ws", b, c, (int)L, 1, n, grad_out, idx, nullptr, grad_points, workspace);
}

extern "C" int pdm_group_concat(void *stream, int b, int n, int m, int c, int nsample,
                                const float *xyz, const float *new_xyz, const float *features,
                                const int *idx, float *out) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c >= 0 && nsample >= 0, PDM_E_BADARG,
                "group_concat: negative size");
    const long long L = (long long)m * nsample;
    if (b == 0 || L == 0) return 0;
    PDM_REQUIRE(xyz && new_xyz && idx && out && (c == 0 || features), PDM_E_BADARG,
                "group_concat: null pointer");
    PDM_REQUIRE(b <= 65535 && 1 + divup(c, GP_CG) <= 65535, PDM_E_TOOLARGE, "group_concat: exceeds grid");
    if (nsample % 4 == 0 && aligned16(idx) && aligned16(out)) {
        dim3 grid(divup(L / 4, GP_THREADS), 1 + divup(c, GP_CG), b);
        hipLaunchKernelGGL(query_group_v4_kernel, grid, dim3(GP_THREADS), 0, as_stream(stream), c, n,
                           m, nsample, xyz, new_xyz, features, idx, out);
    } else {
        dim3 grid(divup(L, GP_THREADS), 1 + divup(c, GP_CG), b);
        hipLaunchKernelGGL(query_group_scalar_kernel, grid, dim3(GP_THREADS), 0, as_stream(stream), c,
                           n, m, nsample, xyz, new_xyz, features, idx, out);
    }
    return check_launch("group_concat");
}

extern "C" int pdm_query_and_group(void *stream, int b, int n, int m, int c, float radius,
                                   int nsample, const float *xyz, const float *new_xyz,
                                   const float *features, int *idx, float *out) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c >= 0 && nsample >= 0, PDM_E_BADARG,
                "query_and_group: negative size");
    const long long L = (long long)m * nsample;
    if (b == 0 || L == 0) return 0;
    PDM_REQUIRE(xyz && new_xyz && idx && out && (c == 0 || features), PDM_E_BADARG,
                "query_and_group: null pointer");
    PDM_REQUIRE(n >= 1, PDM_E_BADARG, "query_and_group: n=%d", n);
    // pointnet2_utils.py:218 — rows of empty balls are zeros
    hipError_t e = hipMemsetAsync(idx, 0, sizeof(int) * (size_t)b * L, as_stream(stream));
    if (e != hipSuccess) {
        set_error("query_and_group: memset failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    int rc = pdm_ball_query(stream, b, n, m, radius, nsample, new_xyz, xyz, idx);
    if (rc != 0) return rc;
    return pdm_group_concat(stream, b, n, m, c, nsample, xyz, new_xyz, features, idx, out);
}

// ---- channels-last QueryAndGroup for the training path -----------------------------------------------------------------
// The autograd path feeds torch/MIOpen convolutions that want NHWC tensors, bf16 under autocast.  pdm_group_concat writes
// the reference's (B, 3+C, M, ns) fp32 layout, which then costs a layout copy and a dtype copy over the largest tensor of
// the step.  This form writes out[b][m][s][3+C] (= a channels-last view of the same logical tensor) in fp32 or bf16
// directly: element e of the output is one thread, rows of the point-major feature table are read contiguously.
// bf16 = round-to-nearest-even of the fp32 value, i.e. exactly what the cast would have produced.
namespace pdm {

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);   // NaN stays NaN (quiet)
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

template <bool BF16>
__global__ __launch_bounds__(256) void group_concat_cl_kernel(long long total, int n, int m, int c, int ns, int ld,
                                                              const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                              const float *__restrict__ feat_pm, const int *__restrict__ idx,
                                                              void *__restrict__ out) {
    const int w = 3 + c;     // ld >= w: row stride of the output; the channels w .. ld - 1 are written as zeros (padding)
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / ld;                // (b, centre, slot)
        const int ch = (int)(e - row * ld);
        const long long bm = row / ns;               // b * m + centre
        const int b = (int)(bm / m);
        const int src = idx[row];
        float v;
        if (ch < 3) v = xyz[((size_t)b * n + src) * 3 + ch] - new_xyz[bm * 3 + ch];   // pointnet2_utils.py:252
        else if (ch >= w) v = 0.0f;
        else v = feat_pm[((size_t)b * n + src) * c + (ch - 3)];
        if constexpr (BF16) static_cast<unsigned short *>(out)[e] = f32_to_bf16_rne(v);
        else static_cast<float *>(out)[e] = v;
    }
}

}  // namespace pdm

// out (B, M, ns, 3+C) fp32 (out_bf16 = 0) or bf16 (1); feat_pm (B, N, C) point-major, may be null when c == 0; idx (B, M, ns).
extern "C" int pdm_group_concat_cl_ld(void *stream, int b, int n, int m, int c, int nsample, const float *xyz, const float *new_xyz,
                                      const float *feat_pm, const int *idx, void *out, int out_bf16, int ld);
namespace pdm {
// (Measured and dropped: a lane group per row with three aligned 16-byte loads per lane and hardware bf16 packing — 113 us per
//  call against ~90: 33- and 65-chunk rows leave half a lane group idle and each lane fetches 48 bytes for 16 written.)
// bf16 rows with ld % 8 == 0 (the padded form): a thread writes EIGHT consecutive channels of one row with one 16-byte store —
// one index / centre lookup and one 64-bit division per 16 bytes instead of per 2 (the element form spent its time there).
template <bool FB>   // FB: the source features are bf16 rows (passed through exactly) instead of fp32
__global__ __launch_bounds__(256) void group_concat_cl8_kernel(long long total8, int n, int m, int c, int ns, int ld,
                                                               const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                               const void *__restrict__ feat_v, const int *__restrict__ idx,
                                                               unsigned short *__restrict__ out) {
    const float *__restrict__ feat_pm = static_cast<const float *>(feat_v);
    const unsigned short *__restrict__ feat_h = static_cast<const unsigned short *>(feat_v);
    const int w = 3 + c, cpr = ld >> 3;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total8; e += (long long)gridDim.x * 256) {
        const long long row = e / cpr;               // (b, centre, slot)
        const int ch0 = (int)(e - row * cpr) << 3;
        const long long bm = row / ns;
        const int b = (int)(bm / m);
        const int src = idx[row];
        const float *__restrict__ f = (!FB && feat_pm) ? feat_pm + ((size_t)b * n + src) * c - 3 : nullptr;
        const unsigned short *__restrict__ fh = (FB && feat_h) ? feat_h + ((size_t)b * n + src) * c - 3 : nullptr;
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int ch = ch0 + t;
            if (ch < 3) v[t] = xyz[((size_t)b * n + src) * 3 + ch] - new_xyz[bm * 3 + ch];
            else if (ch < w) v[t] = FB ? __uint_as_float((unsigned)fh[ch] << 16) : f[ch];
            else v[t] = 0.0f;
        }
        uint4 o;
        o.x = (unsigned)f32_to_bf16_rne(v[0]) | ((unsigned)f32_to_bf16_rne(v[1]) << 16);
        o.y = (unsigned)f32_to_bf16_rne(v[2]) | ((unsigned)f32_to_bf16_rne(v[3]) << 16);
        o.z = (unsigned)f32_to_bf16_rne(v[4]) | ((unsigned)f32_to_bf16_rne(v[5]) << 16);
        o.w = (unsigned)f32_to_bf16_rne(v[6]) | ((unsigned)f32_to_bf16_rne(v[7]) << 16);
        *reinterpret_cast<uint4 *>(out + e * 8) = o;
    }
}
}  // namespace pdm

extern "C" int pdm_group_concat_cl(void *stream, int b, int n, int m, int c, int nsample, const float *xyz,
                                   const float *new_xyz, const float *feat_pm, const int *idx, void *out, int out_bf16) {
    return pdm_group_concat_cl_ld(stream, b, n, m, c, nsample, xyz, new_xyz, feat_pm, idx, out, out_bf16, 3 + c);
}

// The same with a row stride: out (B, M, ns, ld), ld >= 3 + C; the channels 3 + C .. ld - 1 are written as ZEROS, so the tensor
// can feed a contraction over ld channels (16-byte rows for the bf16 MFMA kernels of train_gemm.hip: ld = round8(3 + C)).
extern "C" int pdm_group_concat_cl_ld_f(void *stream, int b, int n, int m, int c, int nsample, const float *xyz, const float *new_xyz,
                                        const void *feat_pm, int feat_bf16, const int *idx, void *out, int out_bf16, int ld);
extern "C" int pdm_group_concat_cl_ld(void *stream, int b, int n, int m, int c, int nsample, const float *xyz, const float *new_xyz,
                                      const float *feat_pm, const int *idx, void *out, int out_bf16, int ld) {
    return pdm_group_concat_cl_ld_f(stream, b, n, m, c, nsample, xyz, new_xyz, feat_pm, 0, idx, out, out_bf16, ld);
}
// the same with the source features as fp32 (feat_bf16 = 0) or bf16 rows (1: only with bf16 output and ld % 8 == 0 — a bf16 feature
// passes into the bf16 result exactly, so this is the fp32 form without the cast of the features in front of it)
extern "C" int pdm_group_concat_cl_ld_f(void *stream, int b, int n, int m, int c, int nsample, const float *xyz, const float *new_xyz,
                                        const void *feat_v, int feat_bf16, const int *idx, void *out, int out_bf16, int ld) {
    const float *feat_pm = static_cast<const float *>(feat_v);
    PDM_REQUIRE(!feat_bf16 || (out_bf16 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0), PDM_E_BADARG,
                "group_concat_cl: bf16 features need the bf16 result with ld a multiple of 8");
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c >= 0 && nsample >= 0 && ld >= 3 + c, PDM_E_BADARG, "group_concat_cl: negative size or ld < 3 + c");
    const long long total = (long long)b * m * nsample * ld;
    if (total == 0) return 0;
    PDM_REQUIRE(xyz && new_xyz && idx && out && (c == 0 || feat_pm), PDM_E_BADARG, "group_concat_cl: null pointer");
    const long long want = (total + 255) / 256;
    const int blocks = (int)(want < 256 * 64 ? want : 256 * 64);
    if (out_bf16 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const long long total8 = total / 8, want8 = (total8 + 255) / 256;
        if (feat_bf16)
            hipLaunchKernelGGL(pdm::group_concat_cl8_kernel<true>, dim3((unsigned)(want8 < 256 * 64 ? want8 : 256 * 64)), dim3(256), 0,
                               pdm::as_stream(stream), total8, n, m, c, nsample, ld, xyz, new_xyz, feat_v, idx, static_cast<unsigned short *>(out));
        else
            hipLaunchKernelGGL(pdm::group_concat_cl8_kernel<false>, dim3((unsigned)(want8 < 256 * 64 ? want8 : 256 * 64)), dim3(256), 0,
                               pdm::as_stream(stream), total8, n, m, c, nsample, ld, xyz, new_xyz, feat_v, idx, static_cast<unsigned short *>(out));
        return pdm::check_launch("group_concat_cl");
    }
    if (out_bf16)
        hipLaunchKernelGGL(pdm::group_concat_cl_kernel<true>, dim3(blocks), dim3(256), 0, pdm::as_stream(stream), total, n, m, c,
                           nsample, ld, xyz, new_xyz, feat_pm, idx, out);
    else
        hipLaunchKernelGGL(pdm::group_concat_cl_kernel<false>, dim3(blocks), dim3(256), 0, pdm::as_stream(stream), total, n, m, c,
                           nsample, ld, xyz, new_xyz, feat_pm, idx, out);
    return pdm::check_launch("group_concat_cl");
}

// Backward of the channels-last form: grad (B, M, ns, 3+C) fp32 or bf16 -> grad_feat_pm (B, N, C) fp32, every element
// written (zero where a point is in no group).  The scatter is inverted per cloud into CSR lists "source point <- grouped
// slots" (counting sort in LDS, one workgroup per cloud); then one wave per source point adds up its slots' rows, lanes over
// channels: every row is read once, contiguously, nothing is atomic.
namespace pdm {

constexpr int GCL_THREADS = 1024;

__global__ __launch_bounds__(GCL_THREADS) void gcl_csr_build_kernel(int ne, int n, const int *__restrict__ idx,
                                                                   int *__restrict__ start_all, int *__restrict__ el_all) {
    extern __shared__ int s_cnt[];   // n counters, then fill cursors
    __shared__ int s_wave[GCL_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int *__restrict__ id = idx + (size_t)b * ne;
    int *__restrict__ start = start_all + (size_t)b * (n + 1);
    int *__restrict__ el = el_all + (size_t)b * ne;
    for (int k = tid; k < n; k += GCL_THREADS) s_cnt[k] = 0;
    __syncthreads();
    for (int e = tid; e < ne; e += GCL_THREADS) {
        const int k = id[e];
        if (k >= 0 && k < n) atomicAdd(&s_cnt[k], 1);
    }
    __syncthreads();
    const int chunk = (n + GCL_THREADS - 1) / GCL_THREADS;
    const int k0 = tid * chunk, k1 = min(k0 + chunk, n);
    int local = 0;
    for (int k = k0; k < k1; ++k) local += s_cnt[k];
    int incl = local;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < wave; ++q) base += s_wave[q];
    int run = base + incl - local;
    for (int k = k0; k < k1; ++k) {
        const int cnt = s_cnt[k];
        start[k] = run;
        s_cnt[k] = run;
        run += cnt;
    }
    if (tid == GCL_THREADS - 1) start[n] = run;
    __syncthreads();
    for (int e = tid; e < ne; e += GCL_THREADS) {
        const int k = id[e];
        if (k < 0 || k >= n) continue;
        el[atomicAdd(&s_cnt[k], 1)] = e;
    }
}

// lpp = lanes per source point (a power of two <= 64, >= min(c, 64) rounded up): with few channels a wave serves 64 / lpp
// points at once (SA1 has ONE feature channel: a wave per point left 63 lanes idle)
template <bool BF16>
__global__ __launch_bounds__(256) void gcl_grad_kernel(int n, int c, int ne, int w, int lpp, const void *__restrict__ grad,
                                                       const int *__restrict__ start_all, const int *__restrict__ el_all,
                                                       float *__restrict__ out_pm) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int ppw = 64 / lpp;                                            // points per wave
    const int k = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ppw + lane / lpp;
    if (k >= n) return;
    const int *__restrict__ start = start_all + (size_t)b * (n + 1);
    const int *__restrict__ el = el_all + (size_t)b * ne;
    const int s = start[k], e = start[k + 1];   // w = row stride of grad (>= 3 + c)
    for (int ch = lane % lpp; ch < c; ch += lpp) {
        float acc = 0.0f;
        for (int p = s; p < e; ++p) {
            const size_t off = ((size_t)b * ne + el[p]) * w + 3 + ch;
            if constexpr (BF16) acc += __uint_as_float((unsigned)static_cast<const unsigned short *>(grad)[off] << 16);
            else acc += static_cast<const float *>(grad)[off];
        }
        out_pm[((size_t)b * n + k) * c + ch] = acc;
    }
}

// The same for bf16 rows of a width that is a multiple of 8 (16-byte aligned rows: the padded training form): a lane owns an 8-byte
// word (four columns, the three coordinate columns included and dropped at the end), so a 208-byte row is ONE wave load instead
// of two 2-byte-per-lane loads, and four occurrences are in flight per step (the scalar form walked a point's occurrences one
// dependent index load + row load at a time: ~1 TB/s over the gradient).  Same summation order per channel.
__global__ __launch_bounds__(256) void gcl_grad_bf16x4_kernel(int n, int c, int ne, int w, int lpp, const uint2 *__restrict__ grad,
                                                              const int *__restrict__ start_all, const int *__restrict__ el_all,
                                                              float *__restrict__ out_pm) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int ppw = 64 / lpp;
    const int k = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ppw + lane / lpp;
    if (k >= n) return;
    const int *__restrict__ start = start_all + (size_t)b * (n + 1);
    const int *__restrict__ el = el_all + (size_t)b * ne;
    const int s = start[k], e = start[k + 1];
    const int words = w >> 2;
    const uint2 *__restrict__ g = grad + (size_t)b * ne * words;
    auto add = [](float (&a)[4], uint2 v) {
        a[0] += __uint_as_float(v.x << 16); a[1] += __uint_as_float(v.x & 0xffff0000u);
        a[2] += __uint_as_float(v.y << 16); a[3] += __uint_as_float(v.y & 0xffff0000u);
    };
    for (int word = lane % lpp; word < words; word += lpp) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        int p = s;
        for (; p + 4 <= e; p += 4) {
            const int e0 = el[p], e1 = el[p + 1], e2 = el[p + 2], e3 = el[p + 3];
            const uint2 v0 = g[(size_t)e0 * words + word], v1 = g[(size_t)e1 * words + word], v2 = g[(size_t)e2 * words + word],
                        v3 = g[(size_t)e3 * words + word];
            add(a, v0); add(a, v1); add(a, v2); add(a, v3);
        }
        for (; p < e; ++p) add(a, g[(size_t)el[p] * words + word]);
        float *__restrict__ o = out_pm + ((size_t)b * n + k) * c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = 4 * word + i - 3;
            if (ch >= 0 && ch < c) o[ch] = a[i];
        }
    }
}

}  // namespace pdm

extern "C" size_t pdm_group_concat_cl_grad_ws_bytes(int b, int n, int m, int nsample) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0) return 0;
    return (size_t)b * ((size_t)(n + 1) + (size_t)m * nsample) * sizeof(int) + 64;
}

// grad (B, M, ns, 3+C) fp32 / bf16 (grad_bf16) -> grad_feat_pm (B, N, C) fp32, fully written.  n <= 16384.
extern "C" int pdm_group_concat_cl_grad_ld(void *stream, int b, int n, int m, int c, int nsample, const void *grad, int grad_bf16, int ld,
                                           const int *idx, float *grad_feat_pm, void *workspace, size_t workspace_bytes);
extern "C" int pdm_group_concat_cl_grad(void *stream, int b, int n, int m, int c, int nsample, const void *grad, int grad_bf16,
                                        const int *idx, float *grad_feat_pm, void *workspace, size_t workspace_bytes) {
    return pdm_group_concat_cl_grad_ld(stream, b, n, m, c, nsample, grad, grad_bf16, 3 + c, idx, grad_feat_pm, workspace, workspace_bytes);
}

// the same for a gradient with row stride ld >= 3 + C (the padded form of pdm_group_concat_cl_ld)
extern "C" int pdm_group_concat_cl_grad_ld(void *stream, int b, int n, int m, int c, int nsample, const void *grad, int grad_bf16, int ld,
                                           const int *idx, float *grad_feat_pm, void *workspace, size_t workspace_bytes) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c >= 0 && nsample >= 0 && ld >= 3 + c, PDM_E_BADARG, "group_concat_cl_grad: negative size or ld < 3 + c");
    if (b == 0 || n == 0 || c == 0) return 0;
    PDM_REQUIRE(n <= 16384 && b <= 65535 && (long long)m * nsample < (1ll << 30), PDM_E_TOOLARGE,
                "group_concat_cl_grad: n=%d (<= 16384), b=%d", n, b);
    PDM_REQUIRE(grad_feat_pm && (m * nsample == 0 || (grad && idx)) && workspace, PDM_E_BADARG, "group_concat_cl_grad: null pointer");
    PDM_REQUIRE(workspace_bytes >= pdm_group_concat_cl_grad_ws_bytes(b, n, m, nsample), PDM_E_BADARG,
                "group_concat_cl_grad: workspace of %zu bytes, need %zu", workspace_bytes,
                pdm_group_concat_cl_grad_ws_bytes(b, n, m, nsample));
    const int ne = m * nsample;
    uintptr_t p = (reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15;
    int *start = reinterpret_cast<int *>(p);
    int *el = start + (size_t)b * (n + 1);
    if ((size_t)n * sizeof(int) + 1024 > 64 * 1024) {   // dynamic + static LDS above the default 64 KB (n near 16384)
        // (the kernel also holds a small static block: dynamic + static must stay within the 160 KB of a CU)
        const int e = pdm::grant_lds(reinterpret_cast<const void *>(&pdm::gcl_csr_build_kernel), 128 * 1024);
        PDM_REQUIRE(e == 0, PDM_E_TOOLARGE, "group_concat_cl_grad: cannot obtain %zu bytes of LDS", (size_t)n * sizeof(int));
    }
    hipLaunchKernelGGL(pdm::gcl_csr_build_kernel, dim3(b), dim3(pdm::GCL_THREADS), (size_t)n * sizeof(int), pdm::as_stream(stream), ne,
                       n, idx, start, el);
    int rc = pdm::check_launch("group_concat_cl_grad(csr)");
    if (rc) return rc;
    if (grad_bf16 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0) {   // 16-byte rows: 8-byte words, four occurrences in flight
        int l4 = 1;
        while (l4 < ld / 4 && l4 < 64) l4 <<= 1;
        const int ppb4 = 4 * (64 / l4);
        const dim3 grid4((unsigned)((n + ppb4 - 1) / ppb4), (unsigned)b);
        hipLaunchKernelGGL(pdm::gcl_grad_bf16x4_kernel, grid4, dim3(256), 0, pdm::as_stream(stream), n, c, ne, ld, l4,
                           static_cast<const uint2 *>(grad), start, el, grad_feat_pm);
        return pdm::check_launch("group_concat_cl_grad");
    }
    int lpp = 1;
    while (lpp < c && lpp < 64) lpp <<= 1;
    const int ppb = 4 * (64 / lpp);                                      // points per workgroup
    const dim3 grid((unsigned)((n + ppb - 1) / ppb), (unsigned)b);
    if (grad_bf16)
        hipLaunchKernelGGL(pdm::gcl_grad_kernel<true>, grid, dim3(256), 0, pdm::as_stream(stream), n, c, ne, ld, lpp, grad, start, el, grad_feat_pm);
    else
        hipLaunchKernelGGL(pdm::gcl_grad_kernel<false>, grid, dim3(256), 0, pdm::as_stream(stream), n, c, ne, ld, lpp, grad, start, el, grad_feat_pm);
    return pdm::check_launch("group_concat_cl_grad");
}
