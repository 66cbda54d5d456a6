// Depthwise 3x3 convolution over a channels-last BEV map, BatchNorm (folded) and ReLU fused:
//   out[b, y, x, c] = relu(sum_{dy,dx in -1..1} w[(dy+1)*3 + dx+1][c] * in[b, y+dy, x+dx, c] + shift[c]),  zero padding.
// The "context learning" stage of the BEV heat-map head (pdm_ssd_amd/dense_heads/pdm_heatmap_head.py; build-defined,
// the reference snapshot holds no source for PDM-SSD's head: SURVEY.md F1).  The PDM neck's grid is channels-last
// storage (B, H, W, C), so a cell is one contiguous row of C floats: a thread owns four channels of one cell, a
// workgroup a run of cells; the nine taps of neighbouring cells hit L1 / L2 (three map rows = 270 KB per cloud), so
// HBM sees the map once in and once out.  Bound: HBM (8 * C bytes per cell).
#include "common.h"

namespace pdm {

__global__ __launch_bounds__(256) void depthwise3x3_cl_kernel(int H, int W, int C4, const float4 *__restrict__ in,
                                                             const float4 *__restrict__ w, const float4 *__restrict__ shift,
                                                             float4 *__restrict__ out, int relu) {
    const int b = blockIdx.y;
    const long long cell0 = (long long)blockIdx.x * (256 / C4);
    const int cq = threadIdx.x % C4, lc = threadIdx.x / C4;
    const long long cell = cell0 + lc;
    if (cell >= (long long)H * W || lc >= 256 / C4) return;
    const int y = (int)(cell / W), x = (int)(cell - (long long)y * W);
    const float4 *__restrict__ img = in + (size_t)b * H * W * C4;
    float4 acc = shift[cq];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const float4 v = img[((size_t)yy * W + xx) * C4 + cq];
            const float4 k = w[((dy + 1) * 3 + dx + 1) * C4 + cq];
            acc.x = fmaf(k.x, v.x, acc.x); acc.y = fmaf(k.y, v.y, acc.y);
            acc.z = fmaf(k.z, v.z, acc.z); acc.w = fmaf(k.w, v.w, acc.w);
        }
    }
    if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    out[((size_t)b * H * W + cell) * C4 + cq] = acc;
}

}  // namespace pdm

// in / out (B, H, W, C) fp32 channels-last (distinct buffers), w (9, C) tap-major with the BatchNorm scale folded in,
// shift (C).  C a multiple of 4, C <= 1024; all pointers 16-byte aligned.
extern "C" int pdm_bev_depthwise3x3(void *stream, int B, int H, int W, int C, const float *in, const float *w,
                                    const float *shift, float *out, int relu) {
    using namespace pdm;
    PDM_REQUIRE(B >= 0 && H >= 0 && W >= 0 && C >= 0, PDM_E_BADARG, "bev_depthwise3x3: negative size");
    if (B == 0 || H == 0 || W == 0 || C == 0) return 0;
    PDM_REQUIRE(C % 4 == 0 && C <= 1024 && B <= 65535, PDM_E_BADARG, "bev_depthwise3x3: C=%d (multiple of 4, <= 1024), B=%d", C, B);
    PDM_REQUIRE(in && w && shift && out && in != out, PDM_E_BADARG, "bev_depthwise3x3: null or aliased pointer");
    PDM_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(shift) |
                  reinterpret_cast<uintptr_t>(out)) & 15) == 0, PDM_E_BADARG, "bev_depthwise3x3: buffers must be 16-byte aligned");
    const int C4 = C / 4, cells_per_wg = 256 / C4 > 0 ? 256 / C4 : 1;
    PDM_REQUIRE(C4 <= 256, PDM_E_BADARG, "bev_depthwise3x3: C=%d", C);
    const long long cells = (long long)H * W;
    hipLaunchKernelGGL(depthwise3x3_cl_kernel, dim3((unsigned)((cells + cells_per_wg - 1) / cells_per_wg), B), dim3(256), 0,
                       as_stream(stream), H, W, C4, reinterpret_cast<const float4 *>(in), reinterpret_cast<const float4 *>(w),
                       reinterpret_cast<const float4 *>(shift), reinterpret_cast<float4 *>(out), relu);
    return check_launch("bev_depthwise3x3");
}
