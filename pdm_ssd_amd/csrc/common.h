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

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int divup(long long a, long long b) { return (int)((a + b - 1) / b); }

// Squared distance with the rounding sequence pinned:
//   d = fma(dz,dz, fma(dy,dy, rn(dx*dx)))
// The translation units are also built with -ffp-contract=off so nothing else is fused.
__device__ __forceinline__ float sqdist(float dx, float dy, float dz) {
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

}  // namespace pdm

#define PDM_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            pdm::set_error(__VA_ARGS__);  \
            return (code);                \
        }                                 \
    } while (0)
