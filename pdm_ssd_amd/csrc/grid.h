// Uniform search grid shared by the grid-accelerated ball query and three_nn (gfx950).
#pragma once
#include "common.h"

namespace pdm {

constexpr int BQG_CAP = 15360;      // max cells per sample (LDS histogram: 60 KB)
constexpr int BQG_BUILD_T = 1024;
constexpr int BQG_HDR = 16;         // floats per sample header

struct GridHdr {                    // lives in the workspace, one per sample (BQG_HDR floats)
    float minx, miny, minz, inv_h;
    int gx, gy, gz, ncells;
};

__device__ __forceinline__ int cell_of(float v, float mn, float inv_h, int g) {
    // monotone non-decreasing in v; NaN maps to cell 0
    const float t = __fmul_rn(v - mn, inv_h);
    return (int)fminf(fmaxf(floorf(t), 0.0f), (float)(g - 1));
}


// per-axis search half-width with rounding margins: every point whose fp32 squared distance is <= r^2
// has |c - x| < r (1 + 2^-21) per axis; R exceeds that by r 2^-10 plus 4x the rounding error of
// forming c +- R in fp32.
__device__ __forceinline__ float search_halfwidth(float c, float absr) {
    return __fmaf_rn(fabsf(c), 2.384185791015625e-07f /* 2^-22 */, absr * 1.0009765625f);
}

// workspace bytes / carve-up shared by both users
static inline size_t grid_workspace_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    // (the second (BQG_CAP + 1)-int table per sample: the running cursors of the split build's scatter kernel)
    return (size_t)b * (BQG_HDR * sizeof(float) + 2 * (size_t)(BQG_CAP + 1) * sizeof(int) + (size_t)n * sizeof(float4)) + 64;
}
struct GridWs {
    float4 *sorted;
    float *hdr;
    int *cell_start;
    int *cursor;
};
static inline GridWs grid_carve(void *workspace, int b, int n) {
    GridWs w;
    uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15;
    w.sorted = reinterpret_cast<float4 *>(base);
    w.hdr = reinterpret_cast<float *>(w.sorted + (size_t)b * n);
    w.cell_start = reinterpret_cast<int *>(w.hdr + (size_t)b * BQG_HDR);
    w.cursor = w.cell_start + (size_t)b * (BQG_CAP + 1);
    return w;
}

// Builds the grid over xyz (B,n,3): cell size h >= 2.02 |radius|, grown until <= max_cells cells.
int launch_grid_build(hipStream_t stream, int b, int n, float radius, int max_cells, const float *xyz,
                      const GridWs &ws);

}  // namespace pdm
