/*
 * oracle/pdm_oracle.c — CPU statement of the Point-Dilation-Mechanism (PDM) neck's scatter.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as pointnet2_oracle.c).
 *
 * PARITY UNPINNED: the reference snapshot contains no PDM code at all (the only
 * description is /root/reference/README.md:12 and the box labels of docs/workflow.svg —
 * "Point Dilation -> Feature Filling -> Projection -> Height Compression").  The arithmetic below
 * is this repository's own written spec (DESIGN.md "PDM spec"); this file is the normative
 * statement of it and the HIP kernels are checked against it.  Conventions borrowed from the
 * reference where they exist:
 *   - grid index = floor((coord - range_min) / cell)           pcdet/models/dense_heads/center_head.py:124-131
 *   - BEV tensor (B, C*D, H, W), channel = c*D + z              pcdet/models/backbones_2d/map_to_bev/height_compression.py:21-23
 *   - flat cell order z, y, x                                   pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:27
 *
 * Spec (per sample b, per sampled point i with position p, features f[0..C), SH coefficients
 * a[0..(L+1)^2), scale sigma):
 *   A10 dilation:  c = floor((p - origin) * inv_cell)  per axis (fp32, one sub + one mul);
 *                  cells g = c + o for o in [-(Kx-1)/2, (Kx-1)/2] x ... ; cells outside the grid are dropped.
 *   A11 filling:   u = centre(g) - p, centre = fma(g + 0.5, cell, origin);  r2 = |u|^2;
 *                  w = (sum_lm a_lm * Y_lm(u / |u|)) * exp(-r2 * inv2s2),  inv2s2 = 1 / (2 sigma^2);
 *                  for r2 == 0 only the l = 0 term is kept.  Real SH basis, positive-constant
 *                  convention (no Condon-Shortley sign), order (l, m) = (0,0),(1,-1),(1,0),(1,1),(2,-2)...
 *   A12 fusion:    grid[b, c, z, y, x] += w * f[c];  wsum[b, z, y, x] += w   (all centres, any order)
 *   A13 height compression: view (B, C, D, H, W) as (B, C*D, H, W).
 *
 * Layouts: xyz (B,P,3); feat (B,P,C); sh (B,P,NSH); inv2s2 (B,P);
 *          grid written as (B, H, W, C*D) [channels-last storage of the logical (B, C*D, H, W)
 *          tensor, inner index q = c*D + z] when layout == 1, or (B, C*D, H, W) contiguous when 0;
 *          wsum (B, H, W, D).
 */
#include <math.h>
#include <stddef.h>

#define PDM_MAX_SH 16

/* Real spherical harmonics up to degree 3 on a unit vector; returns count written. */
static int pdm_sh_basis(int degree, float x, float y, float z, float *Y) {
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

/* weight of point (p, a, inv2s2) at cell centre ctr */
static float pdm_weight(int degree, const float *a, float inv2s2, float ux, float uy, float uz) {
    const float r2 = fmaf(uz, uz, fmaf(uy, uy, ux * ux));
    float Y[PDM_MAX_SH];
    float s;
    if (r2 > 0.0f) {
        const float inv = 1.0f / sqrtf(r2);
        const int nsh = pdm_sh_basis(degree, ux * inv, uy * inv, uz * inv, Y);
        s = 0.0f;
        for (int t = 0; t < nsh; ++t) s = fmaf(a[t], Y[t], s);
    } else {
        s = a[0] * 0.28209479177387814f;
    }
    return s * expf(-r2 * inv2s2);
}

/*
 * grid and wsum must be zero-filled by the caller.
 * origin[3], cell[3], inv_cell[3] (fp32, inv_cell supplied by the host so both sides use the same bits).
 * dims: W (x), H (y), D (z).  K: kx, ky, kz (odd).
 */
int oracle_pdm_scatter(int B, int P, int C, int degree, const float *xyz, const float *feat,
                       const float *sh, const float *inv2s2, const float *origin, const float *cell,
                       const float *inv_cell, int W, int H, int D, int kx, int ky, int kz, int layout,
                       float *grid, float *wsum) {
    if (degree < 0 || degree > 3) return 1;
    if (!(kx & 1) || !(ky & 1) || !(kz & 1)) return 2;
    const int nsh = (degree + 1) * (degree + 1);
    const int hx = kx / 2, hy = ky / 2, hz = kz / 2;
    const size_t CD = (size_t)C * D;
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < P; ++i) {
            const float *p = xyz + ((size_t)b * P + i) * 3;
            const float *f = feat + ((size_t)b * P + i) * C;
            const float *a = sh + ((size_t)b * P + i) * nsh;
            const float is2 = inv2s2[(size_t)b * P + i];
            if (!(p[0] == p[0]) || !(p[1] == p[1]) || !(p[2] == p[2])) continue; /* NaN point */
            const float fx = floorf((p[0] - origin[0]) * inv_cell[0]);
            const float fy = floorf((p[1] - origin[1]) * inv_cell[1]);
            const float fz = floorf((p[2] - origin[2]) * inv_cell[2]);
            /* far-outside points cannot reach the grid; also keeps the int conversion defined */
            if (fx < -(float)kx || fx > (float)(W + kx) || fy < -(float)ky || fy > (float)(H + ky) ||
                fz < -(float)kz || fz > (float)(D + kz))
                continue;
            const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
            for (int oz = -hz; oz <= hz; ++oz) {
                const int gz = cz + oz;
                if (gz < 0 || gz >= D) continue;
                for (int oy = -hy; oy <= hy; ++oy) {
                    const int gy = cy + oy;
                    if (gy < 0 || gy >= H) continue;
                    for (int ox = -hx; ox <= hx; ++ox) {
                        const int gx = cx + ox;
                        if (gx < 0 || gx >= W) continue;
                        const float ux = fmaf((float)gx + 0.5f, cell[0], origin[0]) - p[0];
                        const float uy = fmaf((float)gy + 0.5f, cell[1], origin[1]) - p[1];
                        const float uz = fmaf((float)gz + 0.5f, cell[2], origin[2]) - p[2];
                        const float w = pdm_weight(degree, a, is2, ux, uy, uz);
                        wsum[(((size_t)b * H + gy) * W + gx) * D + gz] += w;
                        if (layout == 1) {
                            float *g = grid + (((size_t)b * H + gy) * W + gx) * CD;
                            for (int c = 0; c < C; ++c) g[(size_t)c * D + gz] += w * f[c];
                        } else {
                            for (int c = 0; c < C; ++c)
                                grid[(((size_t)b * CD + (size_t)c * D + gz) * H + gy) * W + gx] += w * f[c];
                        }
                    }
                }
            }
        }
    }
    return 0;
}

/* optional A12 normalisation: grid[.., z, ..] /= wsum where |wsum| > eps (cells never touched stay 0) */
int oracle_pdm_normalize(int B, int C, int W, int H, int D, int layout, float eps, float *grid,
                         const float *wsum) {
    const size_t CD = (size_t)C * D;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int z = 0; z < D; ++z) {
                    const float ws = wsum[(((size_t)b * H + y) * W + x) * D + z];
                    if (!(fabsf(ws) > eps)) continue;
                    const float inv = 1.0f / ws;
                    for (int c = 0; c < C; ++c) {
                        const size_t q = (size_t)c * D + z;
                        const size_t off = layout == 1 ? (((size_t)b * H + y) * W + x) * CD + q
                                                       : (((size_t)b * CD + q) * H + y) * W + x;
                        grid[off] *= inv;
                    }
                }
    return 0;
}

/*
 * Backward of oracle_pdm_scatter w.r.t. feat, sh and inv2s2 (xyz gets no gradient: the cell
 * assignment is piecewise constant and the reference family treats sampled coordinates as data).
 *   w = S * G;  dL/df[c] = sum_g w_g * dgrid[g,c];  dL/dw_g = sum_c dgrid[g,c] * f[c] (+ dwsum[g] if given)
 *   dL/da_t = sum_g dw_g * G_g * Y_t(u_g);   dL/dinv2s2 = sum_g dw_g * w_g * (-r2_g)
 */
int oracle_pdm_scatter_grad(int B, int P, int C, int degree, const float *xyz, const float *feat,
                            const float *sh, const float *inv2s2, const float *origin,
                            const float *cell, const float *inv_cell, int W, int H, int D, int kx,
                            int ky, int kz, int layout, const float *dgrid, const float *dwsum,
                            float *dfeat, float *dsh, float *dinv2s2) {
    if (degree < 0 || degree > 3) return 1;
    const int nsh = (degree + 1) * (degree + 1);
    const int hx = kx / 2, hy = ky / 2, hz = kz / 2;
    const size_t CD = (size_t)C * D;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < P; ++i) {
            const size_t pi = (size_t)b * P + i;
            const float *p = xyz + pi * 3;
            const float *f = feat + pi * C;
            const float *a = sh + pi * nsh;
            const float is2 = inv2s2[pi];
            float *df = dfeat + pi * C;
            float *da = dsh + pi * nsh;
            for (int c = 0; c < C; ++c) df[c] = 0.0f;
            for (int t = 0; t < nsh; ++t) da[t] = 0.0f;
            dinv2s2[pi] = 0.0f;
            if (!(p[0] == p[0]) || !(p[1] == p[1]) || !(p[2] == p[2])) continue;
            const float fx = floorf((p[0] - origin[0]) * inv_cell[0]);
            const float fy = floorf((p[1] - origin[1]) * inv_cell[1]);
            const float fz = floorf((p[2] - origin[2]) * inv_cell[2]);
            if (fx < -(float)kx || fx > (float)(W + kx) || fy < -(float)ky || fy > (float)(H + ky) ||
                fz < -(float)kz || fz > (float)(D + kz))
                continue;
            const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
            double acc_is2 = 0.0;
            for (int oz = -hz; oz <= hz; ++oz) {
                const int gz = cz + oz;
                if (gz < 0 || gz >= D) continue;
                for (int oy = -hy; oy <= hy; ++oy) {
                    const int gy = cy + oy;
                    if (gy < 0 || gy >= H) continue;
                    for (int ox = -hx; ox <= hx; ++ox) {
                        const int gx = cx + ox;
                        if (gx < 0 || gx >= W) continue;
                        const float ux = fmaf((float)gx + 0.5f, cell[0], origin[0]) - p[0];
                        const float uy = fmaf((float)gy + 0.5f, cell[1], origin[1]) - p[1];
                        const float uz = fmaf((float)gz + 0.5f, cell[2], origin[2]) - p[2];
                        const float r2 = fmaf(uz, uz, fmaf(uy, uy, ux * ux));
                        float Y[PDM_MAX_SH];
                        int ny = 1;
                        Y[0] = 0.28209479177387814f;
                        if (r2 > 0.0f) {
                            const float inv = 1.0f / sqrtf(r2);
                            ny = pdm_sh_basis(degree, ux * inv, uy * inv, uz * inv, Y);
                        }
                        float s = 0.0f;
                        for (int t = 0; t < ny; ++t) s = fmaf(a[t], Y[t], s);
                        const float G = expf(-r2 * is2);
                        const float w = s * G;
                        float dw = dwsum ? dwsum[(((size_t)b * H + gy) * W + gx) * D + gz] : 0.0f;
                        for (int c = 0; c < C; ++c) {
                            const size_t q = (size_t)c * D + gz;
                            const size_t off = layout == 1 ? (((size_t)b * H + gy) * W + gx) * CD + q
                                                           : (((size_t)b * CD + q) * H + gy) * W + gx;
                            const float dg = dgrid[off];
                            df[c] += w * dg;
                            dw += dg * f[c];
                        }
                        for (int t = 0; t < ny; ++t) da[t] += dw * G * Y[t];
                        acc_is2 += (double)dw * (double)w * (double)(-r2);
                    }
                }
            }
            dinv2s2[pi] = (float)acc_is2;
        }
    return 0;
}
