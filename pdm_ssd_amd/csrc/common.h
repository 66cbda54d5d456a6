// Shared device/host helpers for libpdmssd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pdmssd_hip.h"

#define PDM_WAVE 64

namespace pdm {

// Thread-local error text behind pdm_last_error().
void set_error(const char *fmt, ...);
int check_launch(const char *what);
// More than 64 KB of dynamic LDS has to be granted per kernel function AND per device (hipFuncSetAttribute acts on the
// CURRENT device's copy of the function; entry points may be called for any device).  Remembers (function, device)
// pairs, thread-safe; returns hipSuccess (0) or the runtime's error code.
int grant_lds(const void *fn, size_t bytes);

// Scatter-add backward through an inverted (CSR) index, shared by three_interpolate and group_points (interpolate.hip)
size_t csr_workspace_bytes(int b, long long ne, int m);
bool csr_form_applies(int b, int row_len, long long ne, int m);
int csr_scatter_grad_launch(void *stream, const char *who, int b, int c, int row_len, int per, int m, const float *grad_out,
                            const int *idx, const float *weight, float *grad_points, void *workspace);

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int divup(long long a, long long b) { return (int)((a + b - 1) / b); }

// Squared distance with the rounding sequence pinned:
//   d = fma(dz,dz, fma(dy,dy, rn(dx*dx)))
// The translation units are also built with -ffp-contract=off so nothing else is fused.
__device__ __forceinline__ float sqdist(float dx, float dy, float dz) {
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

// One radix-select round, run by the FIRST WAVE of the workgroup (all 64 lanes): the first of 256 histogram bins at
// which the running count reaches `remaining` (>= 1), and the count in front of it.  Four bins per lane, one wave scan
// (a single thread walking the bins is a chain of 256 dependent LDS reads, ~10 us per round).
__device__ __forceinline__ void radix_pick256(const int *hist, int remaining, int *digit, int *before) {
    const int lane = threadIdx.x & 63;
    const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const int sum = h0 + h1 + h2 + h3;
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    const int excl = incl - sum;
    const unsigned long long hit = __ballot(excl < remaining && incl >= remaining);
    const int src = hit ? __ffsll((long long)hit) - 1 : 63;   // no lane reaches it only if remaining > total: last bins
    if (lane == src) {
        int acc = excl, d = 4 * lane;
        if (acc + h0 < remaining) { acc += h0; ++d;
            if (acc + h1 < remaining) { acc += h1; ++d;
                if (acc + h2 < remaining) { acc += h2; ++d; } } }
        *digit = d;
        *before = acc;
    }
}

// ---- ragged ("stacked") batches: per-sample counts -> LDS prefix tables (stack_ops.hip, vector_pool.hip) -------
constexpr int ST_MAXB = 1024;   // samples per call held as LDS prefix tables

// prefix[k] = sum of cnt[0..k) for k = 0..B, built once per workgroup
__device__ __forceinline__ void build_prefix(int B, const int *__restrict__ cnt, int *prefix) {
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = 0; k < B; ++k) { prefix[k] = acc; acc += cnt[k]; }
        prefix[B] = acc;
    }
}
// the reference's linear scan (ball_query_gpu.cu:27-32): the last sample absorbs elements past the total
__device__ __forceinline__ int sample_of(int i, int B, const int *prefix) {
    int bs = 0;
    for (int k = 1; k < B; ++k) {
        if (i < prefix[k]) break;
        bs = k;
    }
    return bs;
}

// both tables of a (centres, points) pair: wave 0 builds one, wave 1 the other; the caller syncs the workgroup
__device__ __forceinline__ void build_prefix_pair(int B, const int *__restrict__ cnt_a, int *prefix_a,
                                                  const int *__restrict__ cnt_b, int *prefix_b) {
    build_prefix(B, cnt_a, prefix_a);
    if (threadIdx.x == 64) {
        int acc = 0;
        for (int k = 0; k < B; ++k) { prefix_b[k] = acc; acc += cnt_b[k]; }
        prefix_b[B] = acc;
    }
}

}  // namespace pdm

#define PDM_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            pdm::set_error(__VA_ARGS__);  \
            return (code);                \
        }                                 \
    } while (0)
