// GPU-side input path: the reference's per-sample `sample_points`
// (pcdet/datasets/processor/data_processor.py:182-212: keep every point beyond 40 m, draw the rest of the quota from
// the near points without replacement, pad short clouds with extra distinct picks, shuffle) and `collate_batch`'s
// batch-index column (pcdet/datasets/dataset.py:237-244) for a whole batch of raw clouds already in HBM, one
// workgroup per cloud.  The reference draws with numpy's global RNG, which no other implementation can replay; the
// draw is therefore DEFINED here by counter-based hashes that the CPU oracle evaluates
// identically, so parity with the oracle is exact:
//   key_s(i) = fmix32(fmix32(seed ^ cloud * 0x9E3779B1 ^ s * 0x7F4A7C15) + i * 0x9E3779B9)     (murmur3 finaliser)
//   "k random points of a set"  := the k members with the smallest (key_1, i)
//   "shuffle"                   := ascending (key_2+copy(entry), i, copy) over the chosen entries.
#include "common.h"

namespace pdm {

constexpr int IP_THREADS = 1024;
constexpr int IP_MAXP = 16384;          // quota per cloud the in-LDS sort handles (128 KB of 8-byte items)

__device__ __host__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ unsigned ip_key(unsigned base, unsigned i) { return fmix32(base + i * 0x9E3779B9u); }

// block-wide exclusive scan of one int per thread (1024 threads); returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int block_scan(int v, int *s_wave, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();                      // s_wave reuse across calls
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < IP_THREADS / 64; ++w) {
        const int x = s_wave[w];
        if (w < wave) base += x;
        tot += x;
    }
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(IP_THREADS) void sample_points_kernel(int B, int P, unsigned seed, int C,
                                                                  const float *__restrict__ raw,
                                                                  const int *__restrict__ counts,
                                                                  float *__restrict__ out, int *__restrict__ choice_out) {
    extern __shared__ unsigned long long s_items[];   // P2 sort items
    __shared__ int s_hist[256];
    __shared__ int s_wave[IP_THREADS / 64];
    __shared__ int s_far, s_digit, s_before;
    const int cloud = blockIdx.x, tid = threadIdx.x;
    long long start = 0;
    for (int k = 0; k < cloud; ++k) start += counts[k];
    const int N = counts[cloud];
    const float *__restrict__ pts = raw + start * C;
    const unsigned base1 = fmix32(seed ^ (unsigned)cloud * 0x9E3779B1u ^ 1u * 0x7F4A7C15u);
    const unsigned base2 = fmix32(seed ^ (unsigned)cloud * 0x9E3779B1u ^ 2u * 0x7F4A7C15u);
    const unsigned base3 = fmix32(seed ^ (unsigned)cloud * 0x9E3779B1u ^ 3u * 0x7F4A7C15u);

    auto is_near = [&](int i) {   // np.linalg.norm over fp32: sqrt(((x*x + y*y) + z*z)), every step rounded
        const float x = pts[(size_t)i * C], y = pts[(size_t)i * C + 1], z = pts[(size_t)i * C + 2];
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
        return __fsqrt_rn(d2) < 40.0f;
    };

    // ---- how many far points -----------------------------------------------------------------------------
    if (tid == 0) s_far = 0;
    __syncthreads();
    {
        int f = 0;
        for (int i = tid; i < N; i += IP_THREADS) f += is_near(i) ? 0 : 1;
        for (int off = 32; off > 0; off >>= 1) f += __shfl_down(f, off, 64);
        if ((tid & 63) == 0 && f) atomicAdd(&s_far, f);
    }
    __syncthreads();
    const int F = s_far;
    // mode 0: P < N, quota exceeds the far points: all far + K = P - F near points;  mode 1: P < N, K = P of all;
    // mode 2: P >= N: every point once + K = P - N extra distinct picks of all
    const int mode = P < N ? (P > F ? 0 : 1) : 2;
    // (the reference's np.random.choice(replace=False) raises when P - N > N; here the surplus rows come out as zeros
    // with choice = -1 and the python wrapper refuses such a call up front)
    const int K = mode == 0 ? P - F : mode == 1 ? P : min(P - N, N);
    auto candidate = [&](int i) { return mode == 0 ? is_near(i) : true; };

    // ---- K-th smallest key among the candidates: 4 rounds of 8-bit radix select ----------------------------------
    unsigned prefix = 0, pmask = 0;
    int remaining = K;                     // rank still to locate inside the current prefix bucket
    if (K > 0) {
        for (int round = 0; round < 4; ++round) {
            const int shift = 24 - 8 * round;
            for (int d = tid; d < 256; d += IP_THREADS) s_hist[d] = 0;
            __syncthreads();
            for (int i = tid; i < N; i += IP_THREADS) {
                if (!candidate(i)) continue;
                const unsigned k = ip_key(base1, (unsigned)i);
                if ((k & pmask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid < 64) radix_pick256(s_hist, remaining, &s_digit, &s_before);
            __syncthreads();
            prefix |= (unsigned)s_digit << shift;
            pmask |= 255u << shift;
            remaining -= s_before;
            __syncthreads();
        }
    }
    const unsigned T = prefix;             // the K-th smallest key; `remaining` = how many keys == T are taken (by index)

    // ---- emission in index order: entry = (point i, copy) -> 8-byte sort item (key2 | i | copy) --------------------
    const int P2 = 1 << (32 - __builtin_clz(max(P, 2) - 1));
    for (int q = tid; q < P2; q += IP_THREADS) s_items[q] = ~0ull;    // padding sorts last
    __syncthreads();
    int emitted = 0, eq_taken = 0;
    for (int c0 = 0; c0 < N; c0 += IP_THREADS) {
        const int i = c0 + tid;
        bool drawn = false, is_eq = false, base_member = false;
        if (i < N) {
            const bool near = is_near(i);
            base_member = mode == 2 || (mode == 0 && !near);           // taken regardless of the draw
            if (K > 0 && (mode != 0 || near)) {
                const unsigned k = ip_key(base1, (unsigned)i);
                drawn = k < T;
                is_eq = k == T;
            }
        }
        int tot_eq;
        const int eq_rank = block_scan(is_eq ? 1 : 0, s_wave, &tot_eq);
        if (is_eq && eq_taken + eq_rank < remaining) drawn = true;
        eq_taken += tot_eq;
        const int n_here = (base_member ? 1 : 0) + (drawn ? 1 : 0);
        int tot;
        const int pos = emitted + block_scan(n_here, s_wave, &tot);
        emitted += tot;
        if (i < N) {
            int p = pos;
            if (base_member) {
                const unsigned long long k2 = ip_key(base2, (unsigned)i);
                if (p < P) s_items[p] = (k2 << 32) | ((unsigned long long)(unsigned)i << 1);
                ++p;
            }
            if (drawn) {
                const bool copy = mode == 2;                            // the extra pick of a point already present
                const unsigned long long k2 = ip_key(copy ? base3 : base2, (unsigned)i);
                if (p < P) s_items[p] = (k2 << 32) | ((unsigned long long)(unsigned)i << 1) | (copy ? 1ull : 0ull);
            }
        }
    }
    __syncthreads();

    // ---- shuffle = bitonic sort of the P2 items ------------------------------------------------------------------
    for (int k = 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < P2; q += IP_THREADS) {
                const int partner = q ^ j;
                if (partner > q) {
                    const unsigned long long a = s_items[q], b = s_items[partner];
                    const bool up = (q & k) == 0;
                    if ((a > b) == up) { s_items[q] = b; s_items[partner] = a; }
                }
            }
            __syncthreads();
        }
    }

    // ---- rows out: [batch index, the C floats of the point] --------------------------------------------------------
    for (int r = tid; r < P; r += IP_THREADS) {
        const unsigned long long it = s_items[r];
        const int i = (int)((unsigned)(it & 0xffffffffull) >> 1);
        float *o = out + ((size_t)cloud * P + r) * (C + 1);
        o[0] = (float)cloud;
        const bool ok = it != ~0ull && i < N;
        for (int c = 0; c < C; ++c) o[1 + c] = ok ? pts[(size_t)i * C + c] : 0.0f;
        if (choice_out) choice_out[(size_t)cloud * P + r] = ok ? i : -1;
    }
}

}  // namespace pdm

using namespace pdm;

extern "C" int pdm_sample_points(void *stream, int B, int num_points, unsigned seed, int C, const float *raw,
                                 const int *counts, float *out, int *choice) {
    PDM_REQUIRE(B >= 0 && num_points >= 1 && C >= 3 && C <= 16, PDM_E_BADARG, "sample_points: B=%d num_points=%d C=%d", B, num_points, C);
    PDM_REQUIRE(num_points <= IP_MAXP, PDM_E_TOOLARGE, "sample_points: num_points=%d exceeds %d", num_points, IP_MAXP);
    if (B == 0) return 0;
    PDM_REQUIRE(raw && counts && out, PDM_E_BADARG, "sample_points: null pointer");
    int p2 = 2;
    while (p2 < num_points) p2 <<= 1;
    const size_t lds = (size_t)p2 * sizeof(unsigned long long);
    if (lds > 64 * 1024) {   // > 64 KB of dynamic LDS has to be granted, per device
        // (the kernel also has ~1.2 KB of static LDS: dynamic + static must stay within the CU's 160 KB)
        const int e = grant_lds(reinterpret_cast<const void *>(&sample_points_kernel), 156 * 1024);
        PDM_REQUIRE(e == 0, PDM_E_TOOLARGE, "sample_points: cannot obtain %zu bytes of LDS: %s", lds, hipGetErrorString((hipError_t)e));
    }
    hipLaunchKernelGGL(sample_points_kernel, dim3(B), dim3(IP_THREADS), lds, as_stream(stream), B, num_points, seed, C, raw, counts,
                       out, choice);
    return check_launch("sample_points");
}
