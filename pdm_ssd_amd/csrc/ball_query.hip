// Ball query for gfx950.
//
// Semantics: /root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/ball_query_gpu.cu:15-51 —
// per centre, the first `nsample` point indices in ASCENDING index order with d2 < radius^2
// (strict), remaining slots padded with the first hit, rows of empty balls left as the caller's
// zeros.  The design is different: the reference gives every centre one thread that walks all N
// points with scalar uncoalesced loads.
//
// Exhaustive kernel (any N): a wave64 owns BQ_CPW centres.  The waves of a workgroup share
// LDS-staged point tiles (coalesced HBM reads; SoA in LDS so lane l reads point base+l without
// bank conflicts); each step a lane holds ONE point and tests it against the wave's centres
// (centre coordinates are wave-uniform), `ballot` gives the hit mask per centre, the running count
// is wave-uniform, hits are written in index order via the mask's prefix popcount, and a centre
// stops consuming points once it has nsample hits.
#include "common.h"

namespace pdm {

constexpr int BQ_WAVES = 4;
constexpr int BQ_TILE = 2048;  // points per LDS tile (24 KB)

// BQ_CPW = centres per wave.  A wave walks its tile steps one after the other (ballot -> count -> branch: a latency
// chain of ~150 cycles per step and centre), so four centres per wave quadruple that chain; with few centres in the
// call (the deeper SA levels: 32 x 256 and 32 x 64) one centre per wave gives four times the waves and a quarter of the
// chain: 13.7 -> measured in profiles/r03_api_ops_device.txt.  Many centres: four per wave share each staged point.
// NW = waves per workgroup: every workgroup stages every tile of its cloud, so with few centres per cloud (SA3: 256 centres
// over 1024 points) 4-wave workgroups read the cloud 64 times over; 16 waves share one staging.
template <int BQ_CPW, int NW = BQ_WAVES>
__global__ __launch_bounds__(NW * 64) void ball_query_wave_kernel(
    int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx) {
    __shared__ float sx[BQ_TILE], sy[BQ_TILE], sz[BQ_TILE];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j0 = (blockIdx.x * NW + wave) * BQ_CPW;  // first centre of this wave
    const float *__restrict__ pts = xyz + (size_t)b * n * 3;
    const unsigned long long below = (1ull << lane) - 1ull;

    float cx[BQ_CPW], cy[BQ_CPW], cz[BQ_CPW];
    int cnt[BQ_CPW], first[BQ_CPW];
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) {
        const bool have = j0 + c < m;
        const float *ctr = new_xyz + ((size_t)b * m + (have ? j0 + c : 0)) * 3;
        cx[c] = ctr[0]; cy[c] = ctr[1]; cz[c] = ctr[2];
        cnt[c] = have ? 0 : nsample;
        first[c] = -1;
    }
    int *__restrict__ out = idx + ((size_t)b * m + j0) * nsample;

    for (int base = 0; base < n; base += BQ_TILE) {
        const int tile = min(BQ_TILE, n - base);
        __syncthreads();  // previous tile fully consumed
        if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(xyz) & 15) == 0) {
            // 16-byte loads, all of a thread's requests in flight before the first LDS write (the scalar loop below was a
            // chain of up to 24 dependent round trips for a tile: most of the 8 us this kernel took on a 1024-point cloud)
            const float4 *__restrict__ p4 = reinterpret_cast<const float4 *>(pts + (size_t)base * 3);
            const int nq = tile * 3 / 4;            // tile % 4 == 0 here (n % 4 == 0, BQ_TILE % 4 == 0)
            constexpr int NU = (BQ_TILE * 3 / 4 + NW * 64 - 1) / (NW * 64);
            float4 v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int q = threadIdx.x + u * NW * 64;
                if (q < nq) v[u] = p4[q];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int q = threadIdx.x + u * NW * 64;
                if (q < nq) {
                    const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = q * 4 + j, pnt = i / 3, comp = i - pnt * 3;
                        (comp == 0 ? sx : comp == 1 ? sy : sz)[pnt] = e[j];
                    }
                }
            }
        } else {
            for (int i = threadIdx.x; i < tile * 3; i += NW * 64) {
                const float v = pts[(size_t)base * 3 + i];
                const int pnt = i / 3, comp = i - pnt * 3;
                (comp == 0 ? sx : comp == 1 ? sy : sz)[pnt] = v;
            }
        }
        __syncthreads();
        bool wave_done = true;
#pragma unroll
        for (int c = 0; c < BQ_CPW; ++c) wave_done = wave_done && cnt[c] >= nsample;
        if (!wave_done) {
            // 256 points per pass: the four 64-point groups are tested first and ONE ballot says whether any of them holds a hit
            // for this centre — usually none does (a ball holds a handful of the cloud's points), and the ordered bookkeeping
            // (ballot per group, prefix popcount, writes) runs only for passes that have one.  The scan is bound by instruction
            // issue (8 waves per SIMD, ~40 instructions per group and centre before): ~12 per group on the common path now.
            for (int s = 0; s < tile; s += 256) {
                float px[4], py[4], pz[4];
                bool in[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = s + u * 64 + lane;
                    in[u] = k < tile;
                    px[u] = in[u] ? sx[k] : 0.f; py[u] = in[u] ? sy[k] : 0.f; pz[u] = in[u] ? sz[k] : 0.f;
                }
                bool all_done = true;
#pragma unroll
                for (int c = 0; c < BQ_CPW; ++c) {
                    if (cnt[c] >= nsample) continue;  // wave-uniform
                    bool hit[4], any = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        hit[u] = in[u] && sqdist(cx[c] - px[u], cy[c] - py[u], cz[c] - pz[u]) < radius2;
                        any = any || hit[u];
                    }
                    if (__ballot(any) != 0ull) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (cnt[c] >= nsample) break;   // wave-uniform: later groups of the pass are not consumed
                            const unsigned long long mask = __ballot(hit[u]);
                            if (mask != 0ull) {
                                if (first[c] < 0) first[c] = base + s + u * 64 + (__ffsll((long long)mask) - 1);
                                const int pos = cnt[c] + __popcll(mask & below);
                                if (hit[u] && pos < nsample) out[(size_t)c * nsample + pos] = base + s + u * 64 + lane;
                                cnt[c] += __popcll(mask);
                            }
                        }
                    }
                    all_done = all_done && cnt[c] >= nsample;
                }
                if (all_done) { wave_done = true; break; }
            }
        }
        // every wave of the block done -> stop staging tiles
        if (__syncthreads_and(wave_done)) break;
    }
    // ball_query_gpu.cu:41-45 — slots beyond the hit count hold the first hit; the row of an EMPTY ball is all zeros: the
    // reference leaves it as its caller zero-filled it (pointnet2_utils.py:218) — written here, so the result does not
    // depend on a fill pass in front of the call (this library's BallQuery allocates without one)
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) {
        if (j0 + c < m && cnt[c] < nsample)
            for (int l = cnt[c] + lane; l < nsample; l += 64) out[(size_t)c * nsample + l] = first[c] >= 0 ? first[c] : 0;
    }
}

}  // namespace pdm

using namespace pdm;

static int g_bq_small_waves = 16;   // exhaustive scan with few centres: waves per workgroup (4 or 16)
extern "C" int pdm_tune_bq_small_waves(int w) { const int old = g_bq_small_waves; if (w == 4 || w == 16) g_bq_small_waves = w; return old; }

extern "C" int pdm_ball_query(void *stream, int b, int n, int m, float radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, PDM_E_BADARG,
                "ball_query: negative size b=%d n=%d m=%d nsample=%d", b, n, m, nsample);
    if (b == 0 || m == 0 || nsample == 0 || n == 0) return 0;
    PDM_REQUIRE(new_xyz && xyz && idx, PDM_E_BADARG, "ball_query: null pointer");
    PDM_REQUIRE(b <= 65535, PDM_E_TOOLARGE, "ball_query: b=%d exceeds grid", b);
    const float radius2 = radius * radius;  // ball_query_gpu.cu:29 (fp32 product)
    if ((long long)b * m <= 32768 && g_bq_small_waves == 16 && m >= 64) {
        dim3 grid(divup(m, 16), b);
        hipLaunchKernelGGL((ball_query_wave_kernel<1, 16>), grid, dim3(16 * 64), 0, as_stream(stream), n,
                           m, radius2, nsample, new_xyz, xyz, idx);
    } else if ((long long)b * m <= 32768) {
        dim3 grid(divup(m, BQ_WAVES), b);
        hipLaunchKernelGGL(ball_query_wave_kernel<1>, grid, dim3(BQ_WAVES * 64), 0, as_stream(stream), n,
                           m, radius2, nsample, new_xyz, xyz, idx);
    } else {
        dim3 grid(divup(m, BQ_WAVES * 4), b);
        hipLaunchKernelGGL(ball_query_wave_kernel<4>, grid, dim3(BQ_WAVES * 64), 0, as_stream(stream), n,
                           m, radius2, nsample, new_xyz, xyz, idx);
    }
    return check_launch("ball_query");
}
