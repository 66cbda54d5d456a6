// Score-ranked sampling: instance-aware down-sampling keeps the npoint points with the
// highest predicted foreground / centre score instead of running FPS (IA-SSD lineage; the sampling code itself is
// absent from the reference snapshot, where the analogous call is torch.topk over per-point scores).  Build-defined,
// total order so that the result is unique and the CPU oracle can match it exactly:
//   rank by score descending on the order-preserving integer image of the float (so -0.0 < +0.0, -inf lowest,
//   NaN of either sign ranks ABOVE +inf, as torch.topk treats it); equal images -> lower index first.
//   idx[b, r] = index of the r-th ranked point, r = 0 .. k-1.
// One workgroup per cloud: 4 rounds of 8-bit radix select find the k-th key, the chosen points are emitted in index
// order (ties with the k-th key by lowest index) and bitonic-sorted in LDS as 8-byte (key, index) items.
#include "common.h"

namespace pdm {

constexpr int TK_THREADS = 1024;
constexpr int TK_MAXK = 16384;   // 128 KB of 8-byte items

// smaller key = higher rank
__device__ __host__ __forceinline__ unsigned topk_key(unsigned bits) {
    if ((bits & 0x7fffffffu) > 0x7f800000u) return 0u;                     // NaN: ranks first
    const unsigned mono = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);   // ascending with the float order
    return ~mono;
}

__device__ __forceinline__ int tk_block_scan(int v, int *s_wave, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < TK_THREADS / 64; ++w) {
        const int x = s_wave[w];
        if (w < wave) base += x;
        tot += x;
    }
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(TK_THREADS) void topk_sampling_kernel(int N, int K, const float *__restrict__ scores,
                                                                  int *__restrict__ idx_out) {
    extern __shared__ unsigned long long s_items[];
    __shared__ int s_hist[256];
    __shared__ int s_wave[TK_THREADS / 64];
    __shared__ int s_digit, s_before;
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const unsigned *__restrict__ sc = reinterpret_cast<const unsigned *>(scores) + (size_t)cloud * N;

    unsigned prefix = 0, pmask = 0;
    int remaining = K;
    for (int round = 0; round < 4; ++round) {
        const int shift = 24 - 8 * round;
        for (int d = tid; d < 256; d += TK_THREADS) s_hist[d] = 0;
        __syncthreads();
        for (int i = tid; i < N; i += TK_THREADS) {
            const unsigned k = topk_key(sc[i]);
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
    const unsigned T = prefix;   // key of the K-th ranked point; `remaining` points with key == T are taken, lowest index first

    const int K2 = 1 << (32 - __builtin_clz(max(K, 2) - 1));
    for (int q = tid; q < K2; q += TK_THREADS) s_items[q] = ~0ull;
    __syncthreads();
    // keys below T are all taken, keys equal to T in index order until `remaining` of them are in: one block scan per
    // 1024 points carries both counts (equal-key count in the high half; N <= 2^31 / 65536 per chunk is trivially met)
    int lt_seen = 0, eq_seen = 0;
    for (int c0 = 0; c0 < N; c0 += TK_THREADS) {
        const int i = c0 + tid;
        unsigned k = 0xffffffffu;
        bool is_lt = false, is_eq = false;
        if (i < N) {
            k = topk_key(sc[i]);
            is_lt = k < T;
            is_eq = k == T;
        }
        int tot;
        const int both = tk_block_scan((is_lt ? 1 : 0) | (is_eq ? 1 << 16 : 0), s_wave, &tot);
        const int lt_rank = both & 0xffff, eq_rank = both >> 16;
        const int eq_before = min(eq_seen + eq_rank, remaining);        // equal-key points taken in front of this one
        const bool take = is_lt || (is_eq && eq_seen + eq_rank < remaining);
        const int pos = lt_seen + lt_rank + eq_before;
        if (take && pos < K) s_items[pos] = ((unsigned long long)k << 32) | (unsigned)i;
        lt_seen += tot & 0xffff;
        eq_seen += tot >> 16;
    }
    __syncthreads();
    for (int k = 2; k <= K2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < K2; q += TK_THREADS) {
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
    for (int r = tid; r < K; r += TK_THREADS) idx_out[(size_t)cloud * K + r] = (int)(unsigned)(s_items[r] & 0xffffffffull);
}

}  // namespace pdm

using namespace pdm;

extern "C" int pdm_topk_sampling(void *stream, int b, int n, int k, const float *scores, int *idx) {
    PDM_REQUIRE(b >= 0 && n >= 0 && k >= 0, PDM_E_BADARG, "topk_sampling: negative size b=%d n=%d k=%d", b, n, k);
    PDM_REQUIRE(k <= n, PDM_E_BADARG, "topk_sampling: k=%d exceeds n=%d", k, n);   // torch.topk raises as well
    PDM_REQUIRE(k <= TK_MAXK, PDM_E_TOOLARGE, "topk_sampling: k=%d exceeds %d", k, TK_MAXK);
    if (b == 0 || k == 0) return 0;
    PDM_REQUIRE(scores && idx, PDM_E_BADARG, "topk_sampling: null pointer");
    int k2 = 2;
    while (k2 < k) k2 <<= 1;
    const size_t lds = (size_t)k2 * sizeof(unsigned long long);
    if (lds > 64 * 1024) {   // > 64 KB of dynamic LDS has to be granted, per device (static LDS comes on top: 156 KB)
        const int e = grant_lds(reinterpret_cast<const void *>(&topk_sampling_kernel), 156 * 1024);
        PDM_REQUIRE(e == 0, PDM_E_TOOLARGE, "topk_sampling: cannot obtain %zu bytes of LDS: %s", lds, hipGetErrorString((hipError_t)e));
    }
    hipLaunchKernelGGL(topk_sampling_kernel, dim3(b), dim3(TK_THREADS), lds, as_stream(stream), n, k, scores, idx);
    return check_launch("topk_sampling");
}
