// PDM neck: point dilation + SH x Gaussian feature filling + multi-centre scatter-add into a BEV
// grid (build-defined spec: DESIGN.md "PDM spec", normative CPU statement oracle/pdm_oracle.c;
// the reference snapshot has no PDM source — SURVEY.md F1).
//
// Forward = atomics-on-HBM.  One wave64 per sampled point:
//   phase A  lane k computes cell k of the point's Kx*Ky*Kz dilation block (index, validity,
//            weight w = SH(u/|u|) * exp(-|u|^2 / 2 sigma^2)) and parks (offset, w) in LDS;
//   phase B  for every valid cell (wave-uniform loop) the 64 lanes add w*f[c] for 64 consecutive
//            channels — with the channels-last grid (layout 1, D == 1) one wave instruction is one
//            256-byte contiguous global_atomic_add_f32 burst, the full-rate shape of the
//            memory-side atomic units (MI355X_MICROARCH.md "Global float atomics").
// Backward = gather (no atomics, deterministic): the same wave re-reads its cells of dgrid.
#include "common.h"

namespace pdm {

constexpr int PDM_WAVES = 4;
constexpr int PDM_MAXK = 512;   // cells per dilation block held in LDS per wave
constexpr int PDM_MAXSH = 16;

__device__ __forceinline__ int sh_basis(int degree, float x, float y, float z, float *Y) {
    Y[0] = 0.28209479177387814f;
    if (degree < 1) return 1;
    Y[1] = 0.4886025119029199f * y;
    Y[2] = 0.4886025119029199f * z;
    Y[3] = 0.4886025119029199f * x;
    if (degree < 2) return 4;
    const float xx = x * x, yy = y * y, zz = z * z;
    Y[4] = 1.0925484305920792f * (x * y);
    Y[5] = 1.0925484305920792f * (y * z);
    Y[6] = 0.31539156525252005f * (3.0f * zz - 1.0f);
    Y[7] = 1.0925484305920792f * (x * z);
    Y[8] = 0.5462742152960396f * (xx - yy);
    if (degree < 3) return 9;
    Y[9] = 0.5900435899266435f * (y * (3.0f * xx - yy));
    Y[10] = 2.890611442640554f * (x * y * z);
    Y[11] = 0.4570457994644658f * (y * (5.0f * zz - 1.0f));
    Y[12] = 0.3731763325901154f * (z * (5.0f * zz - 3.0f));
    Y[13] = 0.4570457994644658f * (x * (5.0f * zz - 1.0f));
    Y[14] = 1.445305721320277f * (z * (xx - yy));
    Y[15] = 0.5900435899266435f * (x * (xx - 3.0f * yy));
    return 16;
}

struct PdmGrid {
    float ox, oy, oz, cx, cy, cz, icx, icy, icz;
    int W, H, D, kx, ky, kz;
};

// Cell k of the dilation block of a point whose base cell is (bx,by,bz): returns false if outside.
__device__ __forceinline__ bool dilate_cell(const PdmGrid &g, int k, int bx, int by, int bz, int &gx,
                                            int &gy, int &gz) {
    const int ox = k % g.kx;
    const int t = k / g.kx;
    const int oy = t % g.ky;
    const int ozz = t / g.ky;
    gx = bx + ox - g.kx / 2;
    gy = by + oy - g.ky / 2;
    gz = bz + ozz - g.kz / 2;
    return gx >= 0 && gx < g.W && gy >= 0 && gy < g.H && gz >= 0 && gz < g.D;
}

// Base cell of a point; false for NaN / far-outside points (cannot reach the grid).
__device__ __forceinline__ bool base_cell(const PdmGrid &g, float px, float py, float pz, int &bx,
                                          int &by, int &bz) {
    if (!(px == px) || !(py == py) || !(pz == pz)) return false;
    const float fx = floorf(__fmul_rn(px - g.ox, g.icx));
    const float fy = floorf(__fmul_rn(py - g.oy, g.icy));
    const float fz = floorf(__fmul_rn(pz - g.oz, g.icz));
    if (fx < -(float)g.kx || fx > (float)(g.W + g.kx) || fy < -(float)g.ky || fy > (float)(g.H + g.ky) ||
        fz < -(float)g.kz || fz > (float)(g.D + g.kz))
        return false;
    bx = (int)fx; by = (int)fy; bz = (int)fz;
    return true;
}

__global__ __launch_bounds__(PDM_WAVES * 64) void pdm_scatter_kernel(
    int B, int P, int C, int degree, PdmGrid g, int layout, const float *__restrict__ xyz,
    const float *__restrict__ feat, const float *__restrict__ sh, const float *__restrict__ inv2s2,
    float *__restrict__ grid, float *__restrict__ wsum) {
    __shared__ float s_w[PDM_WAVES][PDM_MAXK];
    __shared__ int s_cell[PDM_WAVES][PDM_MAXK];  // (gy*W + gx) or -1
    __shared__ int s_z[PDM_WAVES][PDM_MAXK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long pi = (long long)blockIdx.x * PDM_WAVES + wave;
    if (pi >= (long long)B * P) return;  // whole wave exits; no block barrier below
    const int b = (int)(pi / P);
    const float px = xyz[pi * 3 + 0], py = xyz[pi * 3 + 1], pz = xyz[pi * 3 + 2];
    int bx, by, bz;
    if (!base_cell(g, px, py, pz, bx, by, bz)) return;
    const int nsh = (degree + 1) * (degree + 1);
    float a[PDM_MAXSH];
#pragma unroll
    for (int t = 0; t < PDM_MAXSH; ++t) a[t] = t < nsh ? sh[pi * nsh + t] : 0.0f;
    const float is2 = inv2s2[pi];
    const int K = g.kx * g.ky * g.kz;
    const int CD = C * g.D;

    // phase A
    for (int k = lane; k < K; k += 64) {
        int gx, gy, gz;
        const bool ok = dilate_cell(g, k, bx, by, bz, gx, gy, gz);
        float w = 0.0f;
        if (ok) {
            const float ux = __fmaf_rn((float)gx + 0.5f, g.cx, g.ox) - px;
            const float uy = __fmaf_rn((float)gy + 0.5f, g.cy, g.oy) - py;
            const float uz = __fmaf_rn((float)gz + 0.5f, g.cz, g.oz) - pz;
            const float r2 = sqdist(ux, uy, uz);
            float s;
            if (r2 > 0.0f) {
                const float inv = 1.0f / sqrtf(r2);
                float Y[PDM_MAXSH];
                const int ny = sh_basis(degree, ux * inv, uy * inv, uz * inv, Y);
                s = 0.0f;
#pragma unroll
                for (int t = 0; t < PDM_MAXSH; ++t)
                    if (t < ny) s = __fmaf_rn(a[t], Y[t], s);
            } else {
                s = a[0] * 0.28209479177387814f;
            }
            w = s * __expf(-r2 * is2);
            atomicAdd(wsum + (((size_t)b * g.H + gy) * g.W + gx) * g.D + gz, w);
        }
        s_w[wave][k] = w;
        s_cell[wave][k] = ok ? gy * g.W + gx : -1;
        s_z[wave][k] = gz;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes are visible to it

    // phase B
    const float *__restrict__ f = feat + pi * C;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const float fc = c < C ? f[c] : 0.0f;
        for (int k = 0; k < K; ++k) {
            const int cell = s_cell[wave][k];
            if (cell < 0) continue;  // wave-uniform
            const float w = s_w[wave][k];
            const int gz = s_z[wave][k];
            if (c < C) {
                const size_t q = (size_t)c * g.D + gz;
                const size_t off = layout == 1 ? ((size_t)b * g.H * g.W + cell) * CD + q
                                               : ((size_t)b * CD + q) * g.H * g.W + cell;
                atomicAdd(grid + off, w * fc);
            }
        }
    }
}

__global__ void pdm_normalize_kernel(long long cells, int C, int W, int H, int D, int layout, float eps,
                                     float *__restrict__ grid, const float *__restrict__ wsum) {
    // one thread per (b,y,x,z,c) element, channel fastest for layout 1
    const long long total = cells * C;
    const int CD = C * D;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        long long cellz, off;
        if (layout == 1) {
            // e = ((b*H+y)*W+x)*CD + (c*D+z)
            const long long col = e / CD;
            const int q = (int)(e - col * CD);
            const int z = q % D;
            cellz = col * D + z;
            off = e;
        } else {
            // e = ((b*CD + q)*H + y)*W + x
            const long long hw = (long long)H * W;
            const long long bq = e / hw;
            const long long yx = e - bq * hw;
            const long long bb = bq / CD;
            const int q = (int)(bq - bb * CD);
            const int z = q % D;
            cellz = (bb * hw + yx) * D + z;
            off = e;
        }
        const float ws = wsum[cellz];
        if (fabsf(ws) > eps) grid[off] *= 1.0f / ws;
    }
}

// The same for the channels-last layout with C * D a multiple of 4: a thread owns 16 bytes (the scalar form above pays a 64-bit
// division per 4-byte element and streamed the 577 MB map at 2.9 TB/s).  Same arithmetic per element: x *= 1 / w where |w| > eps.
__global__ __launch_bounds__(256) void pdm_normalize_cl4_kernel(long long nvec, int CD4, int D, float eps, float4 *__restrict__ grid,
                                                                const float *__restrict__ wsum) {
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long long)gridDim.x * blockDim.x) {
        const long long col = v / CD4;                       // (b, y, x) cell
        float4 x = grid[v];
        if (D == 1) {
            const float ws = wsum[col];
            if (fabsf(ws) > eps) { const float inv = 1.0f / ws; x.x *= inv; x.y *= inv; x.z *= inv; x.w *= inv; grid[v] = x; }
        } else {
            const int q0 = (int)(v - col * CD4) * 4;         // first of the four inner indices c * D + z
            float *e = reinterpret_cast<float *>(&x);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ws = wsum[col * D + (q0 + i) % D];
                if (fabsf(ws) > eps) e[i] *= 1.0f / ws;
            }
            grid[v] = x;
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Backward (gather): dfeat (B,P,C), dsh (B,P,nsh), dinv2s2 (B,P) fully written.
// (five waves per SIMD: 96 VGPRs, one spill — the kernel is a latency chain, 442 -> 419 us for the neck's backward; six spill 16)
__global__ __launch_bounds__(PDM_WAVES * 64, 5) void pdm_scatter_grad_kernel(
    int B, int P, int C, int degree, PdmGrid g, int layout, const float *__restrict__ xyz,
    const float *__restrict__ feat, const float *__restrict__ sh, const float *__restrict__ inv2s2,
    const float *__restrict__ dgrid, const float *__restrict__ dwsum, float *__restrict__ dfeat,
    float *__restrict__ dsh, float *__restrict__ dinv2s2, const float *__restrict__ nwsum, float neps) {
    // per wave, K entries each (dynamic LDS sized to the window, so that residency is not capped by the largest one):
    // weight, weight gradient, 1 / wsum of the cell (1 without nwsum), cell index (or -1), height bin
    extern __shared__ __attribute__((aligned(16))) float sg_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Kw = g.kx * g.ky * g.kz;
    float *s_w = sg_lds + (size_t)wave * 5 * Kw, *s_dw = s_w + Kw, *s_inv = s_dw + Kw;
    int *s_cell = reinterpret_cast<int *>(s_inv + Kw), *s_z = s_cell + Kw;
    const long long pi = (long long)blockIdx.x * PDM_WAVES + wave;
    if (pi >= (long long)B * P) return;
    const int b = (int)(pi / P);
    const int nsh = (degree + 1) * (degree + 1);
    const float px = xyz[pi * 3 + 0], py = xyz[pi * 3 + 1], pz = xyz[pi * 3 + 2];
    int bx, by, bz;
    const bool live = base_cell(g, px, py, pz, bx, by, bz);
    if (!live) {
        for (int c = lane; c < C; c += 64) dfeat[pi * C + c] = 0.0f;
        if (lane < nsh) dsh[pi * nsh + lane] = 0.0f;
        if (lane == 0) dinv2s2[pi] = 0.0f;
        return;
    }
    float a[PDM_MAXSH];
#pragma unroll
    for (int t = 0; t < PDM_MAXSH; ++t) a[t] = t < nsh ? sh[pi * nsh + t] : 0.0f;
    const float is2 = inv2s2[pi];
    const int K = g.kx * g.ky * g.kz;
    const int CD = C * g.D;

    // weights per cell (as forward phase A, no atomics)
    for (int k = lane; k < K; k += 64) {
        int gx, gy, gz;
        const bool ok = dilate_cell(g, k, bx, by, bz, gx, gy, gz);
        float w = 0.0f, dw0 = 0.0f, inv = 1.0f;
        if (ok) {
            const float ux = __fmaf_rn((float)gx + 0.5f, g.cx, g.ox) - px;
            const float uy = __fmaf_rn((float)gy + 0.5f, g.cy, g.oy) - py;
            const float uz = __fmaf_rn((float)gz + 0.5f, g.cz, g.oz) - pz;
            const float r2 = sqdist(ux, uy, uz);
            float s;
            if (r2 > 0.0f) {
                const float inv = 1.0f / sqrtf(r2);
                float Y[PDM_MAXSH];
                const int ny = sh_basis(degree, ux * inv, uy * inv, uz * inv, Y);
                s = 0.0f;
#pragma unroll
                for (int t = 0; t < PDM_MAXSH; ++t)
                    if (t < ny) s = __fmaf_rn(a[t], Y[t], s);
            } else {
                s = a[0] * 0.28209479177387814f;
            }
            w = s * __expf(-r2 * is2);
            const size_t ci = (((size_t)b * g.H + gy) * g.W + gx) * g.D + gz;
            if (dwsum) dw0 = dwsum[ci];
            if (nwsum) { const float ws = nwsum[ci]; if (fabsf(ws) > neps) inv = 1.0f / ws; }
        }
        s_w[k] = w;
        s_inv[k] = inv;
        s_dw[k] = dw0;
        s_cell[k] = ok ? gy * g.W + gx : -1;
        s_z[k] = gz;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);

    // dfeat[c] = sum_k w_k dgrid[k,c];  dw_k += sum_c dgrid[k,c] f[c]
    const float *__restrict__ f = feat + pi * C;
    if (layout == 1 && g.D == 1 && (C & 1) == 0) {
        // channels-last map, one height bin: a lane owns the channel pair (2 lane, 2 lane + 1) — one 8-byte load per cell covers
        // 128 channels — and the rows of fourteen cells are requested before the first is consumed.  (One 4-byte load at a time,
        // each consumed before the next was requested, made a wave a chain of 2 K round trips: 0.68 ms per step at bs = 32.)
        const size_t HW = (size_t)g.H * g.W;
        for (int c0 = 0; c0 < C; c0 += 128) {
            const int c = c0 + 2 * lane;
            const bool on = c < C;
            const float2 fc = on ? *reinterpret_cast<const float2 *>(f + c) : make_float2(0.0f, 0.0f);
            float2 acc = make_float2(0.0f, 0.0f);
            constexpr int U = 14;
            for (int k0 = 0; k0 < K; k0 += U) {
                float2 dg[U];
                int cells[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int k = k0 + j;
                    cells[j] = k < K ? s_cell[k] : -1;
                    dg[j] = make_float2(0.0f, 0.0f);
                    if (cells[j] >= 0 && on) dg[j] = *reinterpret_cast<const float2 *>(dgrid + ((size_t)b * HW + cells[j]) * C + c);
                }
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    if (cells[j] < 0) continue;         // wave-uniform
                    const float inv = s_inv[k0 + j], w = s_w[k0 + j] * inv;     // dgrid is the gradient of grid / wsum with nwsum
                    acc.x += w * dg[j].x;
                    acc.y += w * dg[j].y;
                    const float part = wave_sum(dg[j].x * fc.x + dg[j].y * fc.y);
                    if (lane == 0) s_dw[k0 + j] += inv * part;
                }
            }
            if (on) *reinterpret_cast<float2 *>(dfeat + pi * C + c) = acc;
        }
    } else
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const float fc = c < C ? f[c] : 0.0f;
        float acc = 0.0f;
        // seven cells (a row of the 7 x 7 window) per pass, their loads issued together: one load at a time, each consumed
        // before the next was requested, made a wave a chain of K round trips (0.68 ms per step at bs = 32)
        constexpr int U = 7;
        for (int k0 = 0; k0 < K; k0 += U) {
            float dg[U];
            int cells[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int k = k0 + j;
                cells[j] = k < K ? s_cell[k] : -1;
                dg[j] = 0.0f;
                if (cells[j] >= 0 && c < C) {
                    const size_t q = (size_t)c * g.D + s_z[k];
                    const size_t off = layout == 1 ? ((size_t)b * g.H * g.W + cells[j]) * CD + q
                                                   : ((size_t)b * CD + q) * g.H * g.W + cells[j];
                    dg[j] = dgrid[off];
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (cells[j] < 0) continue;             // wave-uniform
                const float inv = s_inv[k0 + j];
                acc += s_w[k0 + j] * inv * dg[j];
                const float part = wave_sum(dg[j] * fc);
                if (lane == 0) s_dw[k0 + j] += inv * part;
            }
        }
        if (c < C) dfeat[pi * C + c] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);

    // dsh_t = sum_k dw_k G_k Y_t(u_k);  dinv2s2 = sum_k dw_k w_k (-r2_k)
    float da[PDM_MAXSH];
#pragma unroll
    for (int t = 0; t < PDM_MAXSH; ++t) da[t] = 0.0f;
    float dis2 = 0.0f;
    for (int k = lane; k < K; k += 64) {
        if (s_cell[k] < 0) continue;
        int gx, gy, gz;
        dilate_cell(g, k, bx, by, bz, gx, gy, gz);
        const float ux = __fmaf_rn((float)gx + 0.5f, g.cx, g.ox) - px;
        const float uy = __fmaf_rn((float)gy + 0.5f, g.cy, g.oy) - py;
        const float uz = __fmaf_rn((float)gz + 0.5f, g.cz, g.oz) - pz;
        const float r2 = sqdist(ux, uy, uz);
        float Y[PDM_MAXSH];
        int ny = 1;
        Y[0] = 0.28209479177387814f;
        if (r2 > 0.0f) {
            const float inv = 1.0f / sqrtf(r2);
            ny = sh_basis(degree, ux * inv, uy * inv, uz * inv, Y);
        }
        const float G = __expf(-r2 * is2);
        const float dw = s_dw[k];
#pragma unroll
        for (int t = 0; t < PDM_MAXSH; ++t)
            if (t < ny) da[t] += dw * G * Y[t];
        dis2 += dw * s_w[k] * (-r2);
    }
#pragma unroll
    for (int t = 0; t < PDM_MAXSH; ++t) {
        if (t < nsh) {
            const float v = wave_sum(da[t]);
            if (lane == 0) dsh[pi * nsh + t] = v;
        }
    }
    dis2 = wave_sum(dis2);
    if (lane == 0) dinv2s2[pi] = dis2;
}

static int check_grid_args(const char *who, int B, int P, int C, int degree, int W, int H, int D,
                           int kx, int ky, int kz, int layout) {
    PDM_REQUIRE(B >= 0 && P >= 0 && C >= 0, PDM_E_BADARG, "%s: negative size", who);
    PDM_REQUIRE(degree >= 0 && degree <= 3, PDM_E_BADARG, "%s: SH degree %d not in [0,3]", who, degree);
    PDM_REQUIRE(W > 0 && H > 0 && D > 0, PDM_E_BADARG, "%s: empty grid %dx%dx%d", who, W, H, D);
    PDM_REQUIRE(kx > 0 && ky > 0 && kz > 0 && (kx & 1) && (ky & 1) && (kz & 1), PDM_E_BADARG,
                "%s: dilation %dx%dx%d must be odd and positive", who, kx, ky, kz);
    PDM_REQUIRE((long long)kx * ky * kz <= PDM_MAXK, PDM_E_TOOLARGE, "%s: dilation block > %d cells", who, PDM_MAXK);
    PDM_REQUIRE(layout == 0 || layout == 1, PDM_E_BADARG, "%s: layout %d", who, layout);
    PDM_REQUIRE((long long)H * W < (1ll << 31), PDM_E_TOOLARGE, "%s: H*W overflows int", who);
    return 0;
}

}  // namespace pdm

using namespace pdm;

extern "C" int pdm_scatter_bev(void *stream, int B, int P, int C, int degree, const float *xyz,
                               const float *feat, const float *sh, const float *inv2s2, float ox,
                               float oy, float oz, float cx, float cy, float cz, float icx, float icy,
                               float icz, int W, int H, int D, int kx, int ky, int kz, int layout,
                               float *grid, float *wsum) {
    int rc = check_grid_args("pdm_scatter_bev", B, P, C, degree, W, H, D, kx, ky, kz, layout);
    if (rc) return rc;
    if ((long long)B * P == 0) return 0;
    PDM_REQUIRE(xyz && sh && inv2s2 && grid && wsum && (C == 0 || feat), PDM_E_BADARG,
                "pdm_scatter_bev: null pointer");
    PdmGrid g{ox, oy, oz, cx, cy, cz, icx, icy, icz, W, H, D, kx, ky, kz};
    const long long blocks = ((long long)B * P + PDM_WAVES - 1) / PDM_WAVES;
    PDM_REQUIRE(blocks < (1ll << 31), PDM_E_TOOLARGE, "pdm_scatter_bev: too many points");
    hipLaunchKernelGGL(pdm_scatter_kernel, dim3((unsigned)blocks), dim3(PDM_WAVES * 64), 0,
                       as_stream(stream), B, P, C, degree, g, layout, xyz, feat, sh, inv2s2, grid, wsum);
    return check_launch("pdm_scatter_bev");
}

extern "C" int pdm_bev_normalize(void *stream, int B, int C, int W, int H, int D, int layout,
                                 float eps, float *grid, const float *wsum) {
    PDM_REQUIRE(B >= 0 && C >= 0 && W > 0 && H > 0 && D > 0, PDM_E_BADARG, "pdm_bev_normalize: bad size");
    PDM_REQUIRE(layout == 0 || layout == 1, PDM_E_BADARG, "pdm_bev_normalize: layout %d", layout);
    const long long cells = (long long)B * H * W * D;
    if (cells == 0 || C == 0) return 0;
    PDM_REQUIRE(grid && wsum, PDM_E_BADARG, "pdm_bev_normalize: null pointer");
    const long long total = cells * C;
    if (layout == 1 && ((long long)C * D) % 4 == 0 && (reinterpret_cast<uintptr_t>(grid) & 15) == 0) {
        const long long nvec = total / 4;
        const int vb = (int)((nvec + 255) / 256 > 16384 ? 16384 : (nvec + 255) / 256);
        hipLaunchKernelGGL(pdm_normalize_cl4_kernel, dim3(vb), dim3(256), 0, as_stream(stream), nvec, C * D / 4, D, eps,
                           reinterpret_cast<float4 *>(grid), wsum);
        return check_launch("pdm_bev_normalize");
    }
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(pdm_normalize_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                       (long long)B * H * W * D, C, W, H, D, layout, eps, grid, wsum);
    return check_launch("pdm_bev_normalize");
}

// Backward of the normalisation y = x / w (where |w| > eps, else y = x) for the channels-last layout, one wave per BEV
// cell: dx = dy / w and dw = -(1/w) sum_c dy_c y_c in the same pass over the cell's C*D contiguous values.
__global__ __launch_bounds__(256) void pdm_normalize_grad_kernel(long long cells_xy, int C, int D, float eps,
                                                                 const float *__restrict__ y, const float *__restrict__ wsum,
                                                                 const float *__restrict__ dy, float *__restrict__ dx,
                                                                 float *__restrict__ dwsum) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    const int CD = C * D;
    for (long long cell = wave0; cell < cells_xy; cell += nwaves) {
        const float *yy = y + cell * CD, *gg = dy + cell * CD;
        float *xx = dx + cell * CD;
        for (int z = 0; z < D; ++z) {
            const float w = wsum[cell * D + z];
            const bool on = fabsf(w) > eps;
            const float inv = on ? 1.0f / w : 1.0f;
            float acc = 0.0f;
            for (int c = lane; c < C; c += 64) {
                const int q = c * D + z;
                const float g = gg[q];
                acc += g * yy[q];
                if (dx) xx[q] = g * inv;
            }
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) dwsum[cell * D + z] = on ? -inv * acc : 0.0f;
        }
    }
}

// The same for D == 1 and C a multiple of 4: LPC lanes (a power of two, <= 64) share a cell with 16-byte loads, a wave holds
// 64 / LPC cells at a time and two such groups are in flight.  (One wave per cell with 4-byte loads streamed y and dy at
// 3.8 TB/s.)  dx may be null.
template <int LPC>
__global__ __launch_bounds__(256) void pdm_normalize_grad_cl4_kernel(long long cells, int C4, float eps, const float4 *__restrict__ y,
                                                                     const float *__restrict__ wsum, const float4 *__restrict__ dy,
                                                                     float4 *__restrict__ dx, float *__restrict__ dwsum) {
    constexpr int CPW = 64 / LPC;                           // cells per wave pass
    const int lane = threadIdx.x & 63, sub = lane & (LPC - 1), slot = lane / LPC;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long c0 = wave0 * (2 * CPW); c0 < cells; c0 += nwaves * (2 * CPW)) {
        float acc[2], inv[2];
        bool on[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long cell = c0 + h * CPW + slot;
            acc[h] = 0.0f; inv[h] = 1.0f; on[h] = false;
            if (cell < cells) {
                const float w = wsum[cell];
                on[h] = fabsf(w) > eps;
                if (on[h]) inv[h] = 1.0f / w;
                for (int q = sub; q < C4; q += LPC) {
                    const float4 g = dy[cell * C4 + q], v = y[cell * C4 + q];
                    acc[h] += g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w;
                    if (dx) dx[cell * C4 + q] = make_float4(g.x * inv[h], g.y * inv[h], g.z * inv[h], g.w * inv[h]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int off = LPC / 2; off >= 1; off >>= 1) acc[h] += __shfl_xor(acc[h], off, 64);
            const long long cell = c0 + h * CPW + slot;
            if (sub == 0 && cell < cells) dwsum[cell] = on[h] ? -inv[h] * acc[h] : 0.0f;
        }
    }
}

extern "C" int pdm_bev_normalize_grad(void *stream, int B, int C, int W, int H, int D, float eps, const float *y,
                                      const float *wsum, const float *dy, float *dx, float *dwsum) {
    PDM_REQUIRE(B >= 0 && C >= 0 && W > 0 && H > 0 && D > 0, PDM_E_BADARG, "pdm_bev_normalize_grad: bad size");
    const long long cells = (long long)B * H * W;
    if (cells == 0 || C == 0) return 0;
    PDM_REQUIRE(y && wsum && dy && dwsum, PDM_E_BADARG, "pdm_bev_normalize_grad: null pointer");   // dx may be null: dwsum only
    const auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (D == 1 && C % 4 == 0 && al16(y) && al16(dy) && al16(dx)) {
        const int C4 = C / 4;
        const int lpc = C4 >= 64 ? 64 : C4 > 16 ? 32 : C4 > 8 ? 16 : 8;
        const long long per_wg = 4 * 2 * (64 / lpc), wgs = (cells + per_wg - 1) / per_wg;
        const dim3 grid((unsigned)(wgs > 8192 ? 8192 : wgs));
#define NG_LAUNCH(L) hipLaunchKernelGGL(pdm_normalize_grad_cl4_kernel<L>, grid, dim3(256), 0, as_stream(stream), cells, C4, eps, \
                                        reinterpret_cast<const float4 *>(y), wsum, reinterpret_cast<const float4 *>(dy),            \
                                        reinterpret_cast<float4 *>(dx), dwsum)
        if (lpc == 64) NG_LAUNCH(64); else if (lpc == 32) NG_LAUNCH(32); else if (lpc == 16) NG_LAUNCH(16); else NG_LAUNCH(8);
#undef NG_LAUNCH
        return check_launch("pdm_bev_normalize_grad");
    }
    const long long want = (cells + 3) / 4;
    hipLaunchKernelGGL(pdm_normalize_grad_kernel, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, as_stream(stream),
                       cells, C, D, eps, y, wsum, dy, dx, dwsum);
    return check_launch("pdm_bev_normalize_grad");
}

static int scatter_bev_grad_launch(void *stream, int B, int P, int C, int degree, const float *xyz,
                                    const float *feat, const float *sh, const float *inv2s2,
                                    float ox, float oy, float oz, float cx, float cy, float cz,
                                    float icx, float icy, float icz, int W, int H, int D, int kx,
                                    int ky, int kz, int layout, const float *dgrid,
                                    const float *dwsum, float *dfeat, float *dsh, float *dinv2s2, const float *nwsum, float neps) {
    int rc = check_grid_args("pdm_scatter_bev_grad", B, P, C, degree, W, H, D, kx, ky, kz, layout);
    if (rc) return rc;
    if ((long long)B * P == 0) return 0;
    PDM_REQUIRE(xyz && sh && inv2s2 && dgrid && dsh && dinv2s2 && (C == 0 || (feat && dfeat)),
                PDM_E_BADARG, "pdm_scatter_bev_grad: null pointer");
    PdmGrid g{ox, oy, oz, cx, cy, cz, icx, icy, icz, W, H, D, kx, ky, kz};
    const long long blocks = ((long long)B * P + PDM_WAVES - 1) / PDM_WAVES;
    PDM_REQUIRE(blocks < (1ll << 31), PDM_E_TOOLARGE, "pdm_scatter_bev_grad: too many points");
    const size_t lds = (size_t)PDM_WAVES * 5 * kx * ky * kz * sizeof(float);          // <= 40 KB at PDM_MAXK cells
    hipLaunchKernelGGL(pdm_scatter_grad_kernel, dim3((unsigned)blocks), dim3(PDM_WAVES * 64), lds,
                       as_stream(stream), B, P, C, degree, g, layout, xyz, feat, sh, inv2s2, dgrid,
                       dwsum, dfeat, dsh, dinv2s2, nwsum, neps);
    return check_launch("pdm_scatter_bev_grad");
}

extern "C" int pdm_scatter_bev_grad(void *stream, int B, int P, int C, int degree, const float *xyz,
                                    const float *feat, const float *sh, const float *inv2s2,
                                    float ox, float oy, float oz, float cx, float cy, float cz,
                                    float icx, float icy, float icz, int W, int H, int D, int kx,
                                    int ky, int kz, int layout, const float *dgrid,
                                    const float *dwsum, float *dfeat, float *dsh, float *dinv2s2) {
    return scatter_bev_grad_launch(stream, B, P, C, degree, xyz, feat, sh, inv2s2, ox, oy, oz, cx, cy, cz, icx, icy, icz, W, H, D, kx, ky, kz,
                                   layout, dgrid, dwsum, dfeat, dsh, dinv2s2, nullptr, 0.0f);
}

// The same gradient for the NORMALISED map y = grid / wsum (where |wsum| > eps): dgrid is dL/dy as it arrives, the division by
// wsum is applied per cell on the way (wsum (B,H,W,D) as the forward left it), and dwsum = pdm_bev_normalize_grad's second output
// (which may then be called with dx = null: no pass that writes dL/dgrid).
extern "C" int pdm_scatter_bev_grad_normalized(void *stream, int B, int P, int C, int degree, const float *xyz,
                                               const float *feat, const float *sh, const float *inv2s2,
                                               float ox, float oy, float oz, float cx, float cy, float cz,
                                               float icx, float icy, float icz, int W, int H, int D, int kx,
                                               int ky, int kz, int layout, const float *dy, const float *wsum, float eps,
                                               const float *dwsum, float *dfeat, float *dsh, float *dinv2s2) {
    PDM_REQUIRE(wsum && dwsum, PDM_E_BADARG, "pdm_scatter_bev_grad_normalized: null pointer");
    return scatter_bev_grad_launch(stream, B, P, C, degree, xyz, feat, sh, inv2s2, ox, oy, oz, cx, cy, cz, icx, icy, icz, W, H, D, kx, ky, kz,
                                   layout, dy, dwsum, dfeat, dsh, dinv2s2, wsum, eps);
}
