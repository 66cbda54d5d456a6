// Ground truth for the fp32 MFMA core used by fused_mlp.hip: what rate does the K-loop body reach with operands
// already in registers (variant 0), with the fragment copies of the real loop (1), and with L1-resident loads (2)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256) void core(int iters, const f4 *__restrict__ src, float *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    f4 acc[4] = {};
    f4 a[4], an[4], b, bn;
    for (int i = 0; i < 4; ++i) { a[i] = src[lane + 64 * i]; an[i] = a[i]; }
    b = src[lane + 256]; bn = b;
    for (int it = 0; it < iters; ++it) {
        if (VAR >= 1) { b = bn; for (int i = 0; i < 4; ++i) a[i] = an[i]; }
        if (VAR == 2) { bn = src[lane + 256 + (it & 1) * 64]; for (int i = 0; i < 4; ++i) an[i] = src[lane + 64 * i + (it & 3) * 320]; }
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b.x, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b.y, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b.z, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b.w, acc[i], 0, 0, 0);
        if (VAR >= 1) { asm volatile("" : "+v"(bn)); for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(an[i])); }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678f) sink[0] = s;
}

// random_data != 0: operands uniform in [-1, 1) instead of zeros (all-zero operands toggle no multiplier bits: the
// clock the power manager grants is not the one a real kernel gets)
extern "C" int run2(int var, int wg_per_cu, int iters, float *ms_out, int random_data);
extern "C" int run(int var, int wg_per_cu, int iters, float *ms_out) { return run2(var, wg_per_cu, iters, ms_out, 0); }
extern "C" int run2(int var, int wg_per_cu, int iters, float *ms_out, int random_data) {
    f4 *src; float *sink;
    hipMalloc(&src, 4096 * sizeof(f4)); hipMalloc(&sink, 4);
    hipMemset(src, 0, 4096 * sizeof(f4));
    if (random_data) {
        static float host[4096 * 4];
        unsigned s = 12345u;
        for (int i = 0; i < 4096 * 4; ++i) { s = s * 1664525u + 1013904223u; host[i] = (float)(s >> 8) * (2.0f / 16777216.0f) - 1.0f; }
        hipMemcpy(src, host, sizeof(host), hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (var == 0) hipLaunchKernelGGL(core<0>, dim3(blocks), dim3(256), 0, 0, iters, src, sink);
        else if (var == 1) hipLaunchKernelGGL(core<1>, dim3(blocks), dim3(256), 0, 0, iters, src, sink);
        else hipLaunchKernelGGL(core<2>, dim3(blocks), dim3(256), 0, 0, iters, src, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    hipEventElapsedTime(ms_out, e0, e1);
    hipFree(src); hipFree(sink);
    return 0;
}
