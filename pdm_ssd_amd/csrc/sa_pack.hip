// Neighbour-list compaction for the fused SA kernels.
//
// ball_query pads a neighbourhood that holds fewer than nsample points with copies of its first hit
// (pointnet2/src/ball_query_gpu.cu:38-46 semantics, restated in ball_query.hip), so the (centre, slot) rows the
// shared MLP sees contain duplicates; max-pool over a multiset equals max-pool over the set.  Sparse LiDAR
// neighbourhoods are the rule (far range: 1-4 real neighbours of 16/32), so the fused kernels take a COMPACTED
// row list instead of the (B, M, nsample) index tensor:
//
//   centre with cnt real slots (cnt = 1 + the last slot that differs from slot 0 — valid for ANY index tensor,
//   duplicates in the middle stay) -> segment of L = 2^ceil(log2 cnt) rows (padding = slot 0 again), segments of
//   equal L stored together, each class starting on a 16-row tile (L = 32: on a tile PAIR), classes in ascending L.
//   A 16-row tile of class L therefore holds 16/L whole centres; the MLP kernels pool over aligned groups of L lanes
//   (the first log2 L stages of the DPP row-max butterfly) and every centre is written exactly once.
//
// Order inside a class is the centre order (stable ranks from a two-level scan, no atomics): results and layout are
// deterministic.  Three small launches: count -> scan (one workgroup) -> fill.
#include "common.h"

namespace pdm {

constexpr int PK_T = 256;      // centres per workgroup in count / fill
constexpr int PK_NCLS = 6;     // L = 1, 2, 4, 8, 16, 32

__device__ __forceinline__ int pack_count_row(const int *__restrict__ row, int ns) {
    const int first = row[0];
    int cnt = 1;
    const int4 *r4 = reinterpret_cast<const int4 *>(row);
    for (int s = 0; s < (ns >> 2); ++s) {
        const int4 v = r4[s];
        if (v.x != first) cnt = 4 * s + 1;
        if (v.y != first) cnt = 4 * s + 2;
        if (v.z != first) cnt = 4 * s + 3;
        if (v.w != first) cnt = 4 * s + 4;
    }
    return cnt;
}
__device__ __forceinline__ int pack_class(int cnt) { return cnt <= 1 ? 0 : 32 - __clz(cnt - 1); }

// One launch serves up to two lists (the two scales of an MSG level): blockIdx.y picks the job.
struct PackJob {
    int ncentres, m, n, ns, nblocks;
    const int *idx;
    unsigned char *cnt;
    int *bh, *meta;
    int2 *pack;
};
struct PackJobs { PackJob j[2]; };

// per-workgroup class histogram of the centres [256 blk, +256)
__global__ __launch_bounds__(PK_T) void sa_pack_count_kernel(PackJobs jobs) {
    const PackJob &J = jobs.j[blockIdx.y];
    if ((int)blockIdx.x >= J.nblocks) return;
    const int ncentres = J.ncentres, ns = J.ns;
    const int *__restrict__ idx = J.idx;
    unsigned char *__restrict__ cnt_out = J.cnt;
    int *__restrict__ bh = J.bh;
    __shared__ int h[PK_NCLS];
    if (threadIdx.x < PK_NCLS) h[threadIdx.x] = 0;
    __syncthreads();
    const int c = blockIdx.x * PK_T + threadIdx.x;
    if (c < ncentres) {
        const int cnt = pack_count_row(idx + (size_t)c * ns, ns);
        cnt_out[c] = (unsigned char)cnt;
        atomicAdd(&h[pack_class(cnt)], 1);   // LDS counter: the sum is order-independent
    }
    __syncthreads();
    if (threadIdx.x < PK_NCLS) bh[blockIdx.x * PK_NCLS + threadIdx.x] = h[threadIdx.x];
}

// One workgroup of PK_NCLS waves: wave k turns bh[:, k] into exclusive prefixes (in place), then thread 0 lays the
// classes out and the workgroup marks the alignment padding rows dead.  meta[0..5] = first row of class k,
// meta[6] = total rows (a multiple of 32), meta[7] = rows of live segments.
__global__ __launch_bounds__(64 * PK_NCLS) void sa_pack_scan_kernel(PackJobs jobs) {
    const PackJob &J = jobs.j[blockIdx.y];
    const int nblocks = J.nblocks;
    int *__restrict__ bh = J.bh;
    int *__restrict__ meta = J.meta;
    int2 *__restrict__ pack = J.pack;
    __shared__ int total[PK_NCLS];
    __shared__ int base[PK_NCLS + 1];
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int run = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 64) {
        const int b = b0 + lane;
        const int v = b < nblocks ? bh[b * PK_NCLS + k] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (b < nblocks) bh[b * PK_NCLS + k] = run + inc - v;
        run += __shfl(inc, 63, 64);
    }
    if (lane == 0) total[k] = run;
    __syncthreads();
    if (threadIdx.x == 0) {
        int r = 0, live = 0;
        for (int c = 0; c < PK_NCLS; ++c) {
            const int align = c == PK_NCLS - 1 ? 32 : 16;
            r = (r + align - 1) / align * align;
            base[c] = r;
            meta[c] = r;
            r += total[c] << c;
            live += total[c] << c;
        }
        r = (r + 31) / 32 * 32;
        base[PK_NCLS] = r;
        meta[6] = r;
        meta[7] = live;
    }
    __syncthreads();
    // dead rows: between the end of class c's segments and the next class start (fewer than 32 each)
    for (int c = 0; c < PK_NCLS; ++c) {
        const int end = base[c] + (total[c] << c);
        for (int r = end + (int)threadIdx.x; r < base[c + 1]; r += blockDim.x) pack[r] = make_int2(0, -1);
    }
}

__global__ __launch_bounds__(PK_T) void sa_pack_fill_kernel(PackJobs jobs) {
    const PackJob &J = jobs.j[blockIdx.y];
    if ((int)blockIdx.x >= J.nblocks) return;
    const int ncentres = J.ncentres, m = J.m, n = J.n, ns = J.ns;
    const int *__restrict__ idx = J.idx;
    const unsigned char *__restrict__ cnt_in = J.cnt;
    const int *__restrict__ bh = J.bh;
    const int *__restrict__ meta = J.meta;
    int2 *__restrict__ pack = J.pack;
    __shared__ int wc[PK_T / 64][PK_NCLS];
    const int c = blockIdx.x * PK_T + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cnt = c < ncentres ? cnt_in[c] : 0;
    const int k = c < ncentres ? pack_class(cnt) : -1;
    int rank = 0;
#pragma unroll
    for (int q = 0; q < PK_NCLS; ++q) {
        const unsigned long long mask = __ballot(k == q);
        if (lane == 0) wc[wv][q] = __popcll(mask);
        if (k == q) rank = __popcll(mask & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (k < 0) return;
    for (int w = 0; w < wv; ++w) rank += wc[w][k];
    rank += bh[blockIdx.x * PK_NCLS + k];
    const int L = 1 << k;
    const int row0 = meta[k] + rank * L;
    const int *row = idx + (size_t)c * ns;
    const int src0 = (c / m) * n;      // rows of the source set are global: b * n + neighbour
    const int first = row[0];
    for (int s = 0; s < L; ++s) pack[row0 + s] = make_int2(src0 + (s < cnt ? row[s] : first), c);
}

}  // namespace pdm

using namespace pdm;

extern "C" size_t pdm_sa_pack_workspace_bytes(int b, int m) {
    const long long nc = (long long)(b > 0 ? b : 0) * (m > 0 ? m : 0);
    const long long nblocks = (nc + PK_T - 1) / PK_T;
    return (size_t)((nc + 15) / 16 * 16 + nblocks * PK_NCLS * (long long)sizeof(int));
}

extern "C" size_t pdm_sa_pack_rows(int b, int m, int nsample) {
    // worst case: every centre keeps nsample rows; + one tile (pair) of alignment per class
    return (size_t)((long long)(b > 0 ? b : 0) * (m > 0 ? m : 0) * nsample + 32 * (PK_NCLS + 1));
}

static int sa_pack_job(const char *who, int b, int n, int m, int nsample, const int *idx, void *workspace, size_t workspace_bytes, int *pack,
                       int *meta, PackJob &J) {
    PDM_REQUIRE(b >= 0 && n >= 1 && m >= 0, PDM_E_BADARG, "%s: bad size b=%d n=%d m=%d", who, b, n, m);
    PDM_REQUIRE(nsample == 16 || nsample == 32, PDM_E_BADARG, "%s: nsample=%d (16 or 32)", who, nsample);
    PDM_REQUIRE(meta && pack, PDM_E_BADARG, "%s: null output", who);
    const long long nc = (long long)b * m;
    PDM_REQUIRE(nc * nsample + 32 * (PK_NCLS + 1) < (1ll << 31) && (long long)b * n < (1ll << 31), PDM_E_TOOLARGE,
                "%s: row numbers overflow 32 bits", who);
    PDM_REQUIRE(nc == 0 || (idx && workspace), PDM_E_BADARG, "%s: null pointer", who);
    PDM_REQUIRE(workspace_bytes >= pdm_sa_pack_workspace_bytes(b, m), PDM_E_BADARG, "%s: workspace %zu < %zu bytes", who,
                workspace_bytes, pdm_sa_pack_workspace_bytes(b, m));
    PDM_REQUIRE((reinterpret_cast<uintptr_t>(idx) & 15) == 0 && (reinterpret_cast<uintptr_t>(pack) & 7) == 0, PDM_E_BADARG,
                "%s: idx must be 16-byte, pack 8-byte aligned", who);
    unsigned char *cnt = static_cast<unsigned char *>(workspace);
    J.ncentres = (int)nc; J.m = m; J.n = n; J.ns = nsample; J.nblocks = (int)((nc + PK_T - 1) / PK_T);
    J.idx = idx; J.cnt = cnt; J.bh = reinterpret_cast<int *>(cnt + (nc + 15) / 16 * 16); J.meta = meta;
    J.pack = reinterpret_cast<int2 *>(pack);
    return 0;
}
static int sa_pack_launch(void *stream, const PackJobs &jobs, int njobs) {
    int nb = 0;
    for (int k = 0; k < njobs; ++k) nb = jobs.j[k].nblocks > nb ? jobs.j[k].nblocks : nb;
    if (nb > 0) hipLaunchKernelGGL(sa_pack_count_kernel, dim3(nb, njobs), dim3(PK_T), 0, as_stream(stream), jobs);
    hipLaunchKernelGGL(sa_pack_scan_kernel, dim3(1, njobs), dim3(64 * PK_NCLS), 0, as_stream(stream), jobs);
    if (nb > 0) hipLaunchKernelGGL(sa_pack_fill_kernel, dim3(nb, njobs), dim3(PK_T), 0, as_stream(stream), jobs);
    return check_launch("sa_pack");
}

extern "C" int pdm_sa_pack(void *stream, int b, int n, int m, int nsample, const int *idx, void *workspace,
                           size_t workspace_bytes, int *pack, int *meta) {
    PackJobs jobs{};
    if (int rc = sa_pack_job("sa_pack", b, n, m, nsample, idx, workspace, workspace_bytes, pack, meta, jobs.j[0])) return rc;
    return sa_pack_launch(stream, jobs, 1);
}

// The neighbour lists of BOTH scales of an MSG level (same b, n, m; nsample / idx / workspace / pack / meta per scale as HOST
// arrays of two) compacted by ONE count -> scan -> fill sequence: three launches for the level instead of six.
extern "C" int pdm_sa_pack_pair(void *stream, int b, int n, int m, const int *nsample, const int *const *idx, void *const *workspace,
                                size_t workspace_bytes, int *const *pack, int *const *meta) {
    PDM_REQUIRE(nsample && idx && workspace && pack && meta, PDM_E_BADARG, "sa_pack_pair: null table");
    PackJobs jobs{};
    for (int k = 0; k < 2; ++k)
        if (int rc = sa_pack_job("sa_pack_pair", b, n, m, nsample[k], idx[k], workspace[k], workspace_bytes, pack[k], meta[k], jobs.j[k])) return rc;
    return sa_pack_launch(stream, jobs, 2);
}
