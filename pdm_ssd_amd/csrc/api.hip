// Error plumbing shared by every entry point of libpdmssd_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <set>
#include <utility>

#include "common.h"

namespace pdm {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Launch-time errors only (bad configuration, missing code object ...); asynchronous faults surface
// at the caller's next synchronisation, as with any HIP launch.  Never exits the process.
int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
}

int grant_lds(const void *fn, size_t bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({fn, dev})) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.insert({fn, dev});
    return (int)e;
}

// Several device-to-device copies in ONE launch (the pipeline's hand-over of ~25 small tensors costs ~10 us per
// hipMemcpyAsync otherwise).  The table travels by value in the kernel arguments.
constexpr int COPY_MAX = 48;
struct CopyTable {
    void *dst[COPY_MAX];
    const void *src[COPY_MAX];
    unsigned long long bytes[COPY_MAX];
    const int *dyn[COPY_MAX];     // optional: only the first *dyn[k] * unit[k] bytes are live (device-side count)
    unsigned unit[COPY_MAX];
};

// VARIANT (pdm_tune_copy_variant): bit 0 = non-temporal stores, bit 1 = eight 16-byte loads in flight per lane (32 KB tiles),
// bit 2 = non-temporal loads.
template <int VARIANT>
__global__ __launch_bounds__(256) void copy_many_kernel(CopyTable t) {
    const int k = blockIdx.y;
    unsigned long long n = t.bytes[k];
    if (t.dyn[k]) {
        const unsigned long long live = (unsigned long long)max(*t.dyn[k], 0) * t.unit[k];
        n = live < n ? live : n;
    }
    char *__restrict__ d = static_cast<char *>(t.dst[k]);
    const char *__restrict__ s = static_cast<const char *>(t.src[k]);
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) {
        const unsigned long long n16 = n >> 4;
        typedef unsigned cp_u4 __attribute__((ext_vector_type(4)));     // (the non-temporal builtins take clang vectors)
        const cp_u4 *__restrict__ s4 = reinterpret_cast<const cp_u4 *>(s);
        cp_u4 *__restrict__ d4 = reinterpret_cast<cp_u4 *>(d);
        // a workgroup walks 16 KB tiles (4 x 256 lanes x 16 bytes, contiguous): four 16-byte loads in flight per lane before the
        // first store (one load per iteration left the memory system a quarter of the requests it needs: 4.7 TB/s on a 1 GiB
        // copy in round 2; pieces a whole grid-stride apart, tried first, were slower still: 4.4 TB/s)
        constexpr int U = (VARIANT & 2) ? 8 : 4;
        constexpr int SH = (VARIANT & 2) ? 11 : 10;
        const unsigned long long ntiles = n16 >> SH;
        for (unsigned long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const unsigned long long i = (tile << SH) + threadIdx.x;
            cp_u4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = (VARIANT & 4) ? __builtin_nontemporal_load(s4 + i + 256 * u) : s4[i + 256 * u];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (VARIANT & 1) __builtin_nontemporal_store(v[u], d4 + i + 256 * u);
                else d4[i + 256 * u] = v[u];
            }
        }
        for (unsigned long long i = (ntiles << SH) + tid; i < n16; i += stride) d4[i] = s4[i];
        for (unsigned long long j = (n16 << 4) + tid; j < n; j += stride) d[j] = s[j];
    } else {
        for (unsigned long long i = tid; i < n; i += stride) d[i] = s[i];
    }
}

}  // namespace pdm

static int copy_many_impl(void *stream, int count, void *const *dst, const void *const *src, const size_t *bytes,
                          const int *const *dyn_count, const unsigned *dyn_unit);
// -1 (default): non-temporal loads and stores (variant 5) for launches whose largest buffer is >= 32 MB — streamed once, nothing
// of it is worth a cache line: 5.4-5.5 -> 6.0 TB/s on a 1 GiB copy (tools/diag/copy_rate.py; the guide's float4 copy: 6.29; torch's
// copy_: 5.05) — and plain accesses below (the pipeline's hand-over: data the next kernels read again); 0..7 force a variant.
static int g_copy_variant = -1;
static int g_copy_max_wg = 8192;
extern "C" int pdm_tune_copy_variant(int v) { const int old = g_copy_variant; g_copy_variant = v < 0 ? -1 : (v & 7); return old; }
extern "C" int pdm_tune_copy_max_wg(int n) { const int old = g_copy_max_wg; if (n > 0) g_copy_max_wg = n; return old; }

extern "C" int pdm_copy_many(void *stream, int count, void *const *dst, const void *const *src, const size_t *bytes) {
    return copy_many_impl(stream, count, dst, src, bytes, nullptr, nullptr);
}

// The same with device-side lengths: where dyn_count[k] is not null only the first *dyn_count[k] * dyn_unit[k] bytes of
// buffer k are copied (read when the kernel runs) — buffers sized for a worst case that hold a device-computed number of
// live rows (the compacted neighbour lists of pdm_sa_pack: count = &meta[6], unit = 8).
extern "C" int pdm_copy_many_dyn(void *stream, int count, void *const *dst, const void *const *src, const size_t *bytes,
                                 const int *const *dyn_count, const unsigned *dyn_unit) {
    PDM_REQUIRE(count == 0 || (dyn_count && dyn_unit), PDM_E_BADARG, "copy_many_dyn: null table");
    return copy_many_impl(stream, count, dst, src, bytes, dyn_count, dyn_unit);
}

static int copy_many_impl(void *stream, int count, void *const *dst, const void *const *src, const size_t *bytes,
                          const int *const *dyn_count, const unsigned *dyn_unit) {
    using namespace pdm;
    PDM_REQUIRE(count >= 0, PDM_E_BADARG, "copy_many: count=%d", count);
    PDM_REQUIRE(count == 0 || (dst && src && bytes), PDM_E_BADARG, "copy_many: null table");
    for (int base = 0; base < count; base += COPY_MAX) {
        CopyTable t;
        int n = 0;
        unsigned long long largest = 0;
        for (int k = base; k < count && n < COPY_MAX; ++k) {
            if (bytes[k] == 0) continue;
            PDM_REQUIRE(dst[k] && src[k], PDM_E_BADARG, "copy_many: null buffer %d", k);
            t.dst[n] = dst[k]; t.src[n] = src[k]; t.bytes[n] = bytes[k];
            t.dyn[n] = dyn_count ? dyn_count[k] : nullptr;
            t.unit[n] = dyn_count ? dyn_unit[k] : 0u;
            largest = bytes[k] > largest ? bytes[k] : largest;
            ++n;
        }
        if (n == 0) continue;
        // a lane moves four 16-byte pieces per pass; up to 8 workgroups per CU
        const unsigned long long want = (largest / 64 + 255) / 256;
        const unsigned gx = (unsigned)(want < 1 ? 1 : want > (unsigned long long)g_copy_max_wg ? (unsigned long long)g_copy_max_wg : want);
        switch (g_copy_variant < 0 ? (largest >= (32ull << 20) ? 5 : 0) : (g_copy_variant & 7)) {
#define PDM_COPY_CASE(V) case V: hipLaunchKernelGGL(copy_many_kernel<V>, dim3(gx, n), dim3(256), 0, as_stream(stream), t); break;
            PDM_COPY_CASE(0) PDM_COPY_CASE(1) PDM_COPY_CASE(2) PDM_COPY_CASE(3) PDM_COPY_CASE(4) PDM_COPY_CASE(5) PDM_COPY_CASE(6) PDM_COPY_CASE(7)
#undef PDM_COPY_CASE
        }
        const int rc = check_launch("copy_many");
        if (rc) return rc;
    }
    return 0;
}

namespace pdm {
__global__ void mark_time_kernel(unsigned long long *slot) { *slot = wall_clock64(); }
}  // namespace pdm

// Diagnostics: writes the device's constant-rate counter (100 MHz on gfx950) to *slot when the stream reaches this point —
// the overlap of the branches of a captured step can be read without a profiler in the way (tools/diag/branch_times.py).
extern "C" int pdm_mark_time(void *stream, unsigned long long *slot) {
    PDM_REQUIRE(slot, PDM_E_BADARG, "mark_time: null slot");
    hipLaunchKernelGGL(pdm::mark_time_kernel, dim3(1), dim3(1), 0, pdm::as_stream(stream), slot);
    return pdm::check_launch("mark_time");
}

extern "C" int pdm_abi_version(void) { return PDM_ABI_VERSION; }
extern "C" const char *pdm_last_error(void) { return pdm::g_err; }
