// Grid-accelerated three_nn for gfx950 — identical (dist2, idx) to the exhaustive scan of
// interpolate.hip / the reference kernel (interpolate_gpu.cu:16-59).
//
// The reference keeps, per unknown point, the three smallest squared distances found while walking the
// known points in ascending index order with a strict '<' cascade; that is the three smallest pairs
// (d, k) in lexicographic order, a definition that does not depend on visiting order.  Here the known
// set is binned into a uniform grid (about two points per cell) and each unknown point (one thread)
// scans an expanding box of cells, comparing candidates by (d, k):
//   1. scan the (2r+1)^3 box around the point's cell, r = 1;
//   2. fewer than three candidates -> double r (until the box is the whole grid);
//   3. otherwise the answer is final only if the box covers [u - D, u + D] per axis, D = sqrt(third best)
//      plus rounding margins; if it does not, rescan exactly that range once (D can only shrink).
// Every point whose fp32 distance is <= the third best lies inside the covered range (same margin
// argument as the grid ball query), so the result is bit-identical to the scan.
#include "grid.h"

namespace pdm {

constexpr int NNG_THREADS = 256;

struct Best3 {
    float d1, d2, d3;
    int i1, i2, i3;
    __device__ __forceinline__ void reset() {
        d1 = d2 = d3 = INFINITY;
        i1 = i2 = i3 = 0;
    }
    // insert candidate (d, k) keeping (d, k)-lexicographic order.  NaN and +inf distances never enter
    // (the reference's "d < 1e40"-initialised cascade rejects them too); an unfilled slot is (inf, 0),
    // which no candidate ties with because k >= 0.
    __device__ __forceinline__ void push(float d, int k) {
        const bool lt1 = d < d1 || (d == d1 && k < i1);
        const bool lt2 = d < d2 || (d == d2 && k < i2);
        const bool lt3 = d < d3 || (d == d3 && k < i3);
        if (lt1) {
            d3 = d2; i3 = i2; d2 = d1; i2 = i1; d1 = d; i1 = k;
        } else if (lt2) {
            d3 = d2; i3 = i2; d2 = d; i2 = k;
        } else if (lt3) {
            d3 = d; i3 = k;
        }
    }
};

__global__ __launch_bounds__(NNG_THREADS) void three_nn_grid_kernel(
    int n, int m, const float *__restrict__ unknown, const float *__restrict__ hdr_all,
    const int *__restrict__ cell_start_all, const float4 *__restrict__ sorted_all,
    float *__restrict__ dist2, int *__restrict__ idx) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * NNG_THREADS + threadIdx.x;
    if (j >= n) return;
    const float *hp = hdr_all + (size_t)b * BQG_HDR;
    const float minx = hp[0], miny = hp[1], minz = hp[2], inv_h = hp[3];
    const int gx = reinterpret_cast<const int *>(hp)[4], gy = reinterpret_cast<const int *>(hp)[5],
              gz = reinterpret_cast<const int *>(hp)[6];
    const int *__restrict__ cell_start = cell_start_all + (size_t)b * (BQG_CAP + 1);
    const float4 *__restrict__ sorted = sorted_all + (size_t)b * m;
    const float *u = unknown + ((size_t)b * n + j) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const int ucx = cell_of(ux, minx, inv_h, gx), ucy = cell_of(uy, miny, inv_h, gy), ucz = cell_of(uz, minz, inv_h, gz);

    Best3 best;
    int x0, x1, y0, y1, z0, z1;
    int r = 1;
    bool exact_range = false;
    for (;;) {
        if (!exact_range) {
            x0 = max(ucx - r, 0); x1 = min(ucx + r, gx - 1);
            y0 = max(ucy - r, 0); y1 = min(ucy + r, gy - 1);
            z0 = max(ucz - r, 0); z1 = min(ucz + r, gz - 1);
        }
        best.reset();
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const int base = (z * gy + y) * gx;
                const int s = cell_start[base + x0], e = cell_start[base + x1 + 1];
                for (int p = s; p < e; ++p) {
                    const float4 q = sorted[p];
                    const float d = sqdist(ux - q.x, uy - q.y, uz - q.z);
                    best.push(d, __float_as_int(q.w));
                }
            }
        const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == gx - 1 && y1 == gy - 1 && z1 == gz - 1;
        if (whole || exact_range) break;
        if (best.d3 < INFINITY) {
            // does the box cover every point that could still beat or tie the third best?
            const float D = sqrtf(best.d3);
            const float hx = search_halfwidth(ux, D), hy = search_halfwidth(uy, D), hz = search_halfwidth(uz, D);
            const int nx0 = cell_of(ux - hx, minx, inv_h, gx), nx1 = cell_of(ux + hx, minx, inv_h, gx);
            const int ny0 = cell_of(uy - hy, miny, inv_h, gy), ny1 = cell_of(uy + hy, miny, inv_h, gy);
            const int nz0 = cell_of(uz - hz, minz, inv_h, gz), nz1 = cell_of(uz + hz, minz, inv_h, gz);
            if (nx0 >= x0 && nx1 <= x1 && ny0 >= y0 && ny1 <= y1 && nz0 >= z0 && nz1 <= z1) break;
            x0 = min(nx0, x0); x1 = max(nx1, x1); y0 = min(ny0, y0); y1 = max(ny1, y1);
            z0 = min(nz0, z0); z1 = max(nz1, z1);
            exact_range = true;  // one more scan over the union; the third best can only improve
        } else {
            r *= 2;
        }
    }
    float *od = dist2 + ((size_t)b * n + j) * 3;
    int *oi = idx + ((size_t)b * n + j) * 3;
    od[0] = best.d1; od[1] = best.d2; od[2] = best.d3;
    oi[0] = best.i1; oi[1] = best.i2; oi[2] = best.i3;
}

}  // namespace pdm

using namespace pdm;

extern "C" size_t pdm_three_nn_grid_workspace_bytes(int b, int m) { return grid_workspace_bytes(b, m); }

// three_nn against a grid of the KNOWN set that pdm_grid_build left in `workspace` (same b, m).
extern "C" int pdm_three_nn_grid_prebuilt(void *stream, int b, int n, int m, const float *unknown, float *dist2, int *idx,
                                          const void *workspace, size_t workspace_bytes) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0, PDM_E_BADARG, "three_nn_grid: negative size");
    if (b == 0 || n == 0) return 0;
    PDM_REQUIRE(m >= 1, PDM_E_BADARG, "three_nn_grid: m=%d (use pdm_three_nn for an empty known set)", m);
    PDM_REQUIRE(unknown && dist2 && idx && workspace, PDM_E_BADARG, "three_nn_grid: null pointer");
    PDM_REQUIRE(b <= 65535, PDM_E_TOOLARGE, "three_nn_grid: b=%d exceeds grid", b);
    PDM_REQUIRE(workspace_bytes >= grid_workspace_bytes(b, m), PDM_E_BADARG,
                "three_nn_grid: workspace of %zu bytes, need %zu", workspace_bytes, grid_workspace_bytes(b, m));
    const GridWs ws = grid_carve(const_cast<void *>(workspace), b, m);
    dim3 grid(divup(n, NNG_THREADS), b);
    hipLaunchKernelGGL(three_nn_grid_kernel, grid, dim3(NNG_THREADS), 0, as_stream(stream), n, m, unknown, ws.hdr,
                       ws.cell_start, ws.sorted, dist2, idx);
    return check_launch("three_nn_grid");
}

extern "C" int pdm_three_nn_grid(void *stream, int b, int n, int m, const float *unknown,
                                 const float *known, float *dist2, int *idx, void *workspace,
                                 size_t workspace_bytes) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0, PDM_E_BADARG, "three_nn_grid: negative size");
    if (b == 0 || n == 0) return 0;
    PDM_REQUIRE(m >= 1, PDM_E_BADARG, "three_nn_grid: m=%d (use pdm_three_nn for an empty known set)", m);
    PDM_REQUIRE(unknown && known && dist2 && idx && workspace, PDM_E_BADARG, "three_nn_grid: null pointer");
    PDM_REQUIRE(b <= 65535, PDM_E_TOOLARGE, "three_nn_grid: b=%d exceeds grid", b);
    PDM_REQUIRE(workspace_bytes >= grid_workspace_bytes(b, m), PDM_E_BADARG,
                "three_nn_grid: workspace of %zu bytes, need %zu", workspace_bytes, grid_workspace_bytes(b, m));
    const GridWs ws = grid_carve(workspace, b, m);
    // about two known points per cell
    int rc = launch_grid_build(as_stream(stream), b, m, 0.0f, m / 2 > 8 ? m / 2 : 8, known, ws);
    if (rc) return rc;
    dim3 grid(divup(n, NNG_THREADS), b);
    hipLaunchKernelGGL(three_nn_grid_kernel, grid, dim3(NNG_THREADS), 0, as_stream(stream), n, m, unknown, ws.hdr,
                       ws.cell_start, ws.sorted, dist2, idx);
    return check_launch("three_nn_grid");
}
