/*
 * oracle/pointnet2_oracle.c — CPU restatement of the reference's pointnet2_batch operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pdm_ssd_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
 * (and as the timed "reference CPU fallback" stand-in: the reference has no CPU path for these
 * operators).
 *
 * Parity status: the reference ships no tests, golden vectors or KATs for this path and its
 * kernels are CUDA (.cu) which cannot be built in this image (no nvcc) — so the kernel-level
 * oracle is "parity unpinned" by reference fixtures.  It is pinned instead by (a) hand-derived
 * known-answer tests in tests/test_oracle_kat.py that follow the cited lines, and (b) module-
 * level fixtures produced by running the reference's own torch modules on top of this oracle
 * (tests/golden/gen_module_fixtures.py).
 *
 * All citations are relative to /root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/.
 *
 * Arithmetic: fp32 throughout, every operation individually rounded (build with
 * -ffp-contract=off, no fast-math).  The squared distance is written in the reference as
 * (a)*(a)+(b)*(b)+(c)*(c) and its rounding sequence is compiler-dependent there; this file
 * offers three sequences (set_dist_mode):
 *   0 PINNED (default, what the HIP kernels implement):
 *         d = fma(dz,dz, fma(dy,dy, rn(dx*dx)))       — nvcc's usual 1 mul + 2 fma chain
 *   1 NONE:  d = rn(rn(rn(dx*dx) + rn(dy*dy)) + rn(dz*dz))  — no contraction
 *   2 HIPCC_DEFAULT: d = rn(fma(dy,dy, rn(dx*dx)) + rn(dz*dz))  — what hipcc -O3 emits unpinned
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static int g_dist_mode = 0;
static int g_threads = 0; /* 0 = leave OpenMP default */

void oracle_set_dist_mode(int mode) { g_dist_mode = mode; }
int oracle_get_dist_mode(void) { return g_dist_mode; }

void oracle_set_threads(int t) {
    g_threads = t;
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#endif
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline float sqdist(float dx, float dy, float dz, int mode) {
    if (mode == 0) {
        float t = dx * dx;
        t = fmaf(dy, dy, t);
        return fmaf(dz, dz, t);
    } else if (mode == 1) {
        float a = dx * dx, b = dy * dy, c = dz * dz;
        float s = a + b;
        return s + c;
    } else {
        float t = fmaf(dy, dy, dx * dx);
        float c = dz * dz;
        return t + c;
    }
}

/* cuda_utils.h:10-14 — opt_n_threads: min(2^floor(log2 n), 1024), at least 1.
 * The reference evaluates log(n)/log(2.0) in double and truncates; do the same. */
int oracle_opt_n_threads(int work_size) {
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}

/* sampling_gpu.cu:100-216 (kernel) + :218-260 (launcher picks block_size = opt_n_threads(n)).
 * Literal simulation of one thread block per sample: per-thread strided scan with strict '>'
 * (:143-144), then the shared-memory tree (:150-211) whose __update (:93-98) keeps the LEFT
 * operand on ties.  temp is caller-initialised (1e10, pointnet2_utils.py:26) and updated in place.
 * block_size <= 0 means "use the reference's rule". */
int oracle_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                   int *idxs, int block_size) {
    if (m <= 0) return 0; /* :108 */
    if (n <= 0) return 0;
    const int S = block_size > 0 ? block_size : oracle_opt_n_threads(n);
    const int mode = g_dist_mode;
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int bi = 0; bi < b; ++bi) {
        const float *ds = dataset + (size_t)bi * n * 3;
        float *tp = temp + (size_t)bi * n;
        int *out = idxs + (size_t)bi * m;
        float *dists = (float *)malloc(sizeof(float) * S);
        int *dists_i = (int *)malloc(sizeof(int) * S);
        if (!dists || !dists_i) { err = 1; free(dists); free(dists_i); continue; }
        int old = 0;
        out[0] = old; /* :118-120 */
        for (int j = 1; j < m; ++j) {
            const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
            for (int tid = 0; tid < S; ++tid) {
                int besti = 0;   /* :126 */
                float best = -1; /* :127 */
                for (int k = tid; k < n; k += S) {
                    const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
                    const float d = sqdist(x2 - x1, y2 - y1, z2 - z1, mode); /* :140 */
                    const float d2 = fminf(d, tp[k]);                        /* :141 */
                    tp[k] = d2;
                    besti = d2 > best ? k : besti; /* :143 */
                    best = d2 > best ? d2 : best;  /* :144 */
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            /* :150-211 — tree over halves S/2, S/4, ... 1; __update(idx1, idx2) at :93-98 */
            for (int half = S / 2; half >= 1; half >>= 1) {
                for (int tid = 0; tid < half; ++tid) {
                    const float v1 = dists[tid], v2 = dists[tid + half];
                    const int i1 = dists_i[tid], i2 = dists_i[tid + half];
                    dists[tid] = v2 > v1 ? v2 : v1; /* max(v1, v2) */
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            }
            old = dists_i[0]; /* :213 */
            out[j] = old;
        }
        free(dists);
        free(dists_i);
    }
    return err;
}

/* sampling_gpu.cu:15-31 — out[b,c,j] = points[b,c,idx[b,j]] */
int oracle_gather_points(int b, int c, int n, int m, const float *points, const int *idx,
                         float *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *p = points + ((size_t)bi * c + ci) * n;
            const int *ix = idx + (size_t)bi * m;
            float *o = out + ((size_t)bi * c + ci) * m;
            for (int j = 0; j < m; ++j) o[j] = p[ix[j]];
        }
    return 0;
}

/* sampling_gpu.cu:53-70 — grad_points[b,c,idx[b,j]] += grad_out[b,c,j] (caller zero-fills,
 * pointnet2_utils.py:67).  Sequential j order here; the reference uses atomicAdd (any order). */
int oracle_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                              float *grad_points) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *g = grad_out + ((size_t)bi * c + ci) * m;
            const int *ix = idx + (size_t)bi * m;
            float *gp = grad_points + ((size_t)bi * c + ci) * n;
            for (int j = 0; j < m; ++j) gp[ix[j]] += g[j];
        }
    return 0;
}

/* ball_query_gpu.cu:15-51 — per centre: ascending scan, strict d2 < radius2 (:40), first hit
 * fills all nsample slots (:41-45), stop after nsample hits (:48).  idx is caller-zeroed
 * (pointnet2_utils.py:218) and left untouched for empty balls.  radius arrives as C float. */
int oracle_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                      const float *xyz, int *idx) {
    const float radius2 = radius * radius; /* :29 */
    const int mode = g_dist_mode;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int j = 0; j < m; ++j) {
            const float *c = new_xyz + ((size_t)bi * m + j) * 3;
            const float *p = xyz + (size_t)bi * n * 3;
            int *o = idx + ((size_t)bi * m + j) * nsample;
            const float nx = c[0], ny = c[1], nz = c[2];
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                const float d2 = sqdist(nx - p[k * 3 + 0], ny - p[k * 3 + 1], nz - p[k * 3 + 2], mode);
                if (d2 < radius2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) o[l] = k;
                    o[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
    return 0;
}

/* group_points_gpu.cu:53-72 — out[b,c,j,s] = points[b,c,idx[b,j,s]] */
int oracle_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                        const int *idx, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *p = points + ((size_t)bi * c + ci) * n;
            const int *ix = idx + (size_t)bi * npoints * nsample;
            float *o = out + ((size_t)bi * c + ci) * npoints * nsample;
            for (size_t t = 0; t < (size_t)npoints * nsample; ++t) o[t] = p[ix[t]];
        }
    return 0;
}

/* group_points_gpu.cu:14-31 — grad_points[b,c,idx[b,j,s]] += grad_out[b,c,j,s]; caller zero-fills
 * (pointnet2_utils.py:190).  Sequential (j,s) order; reference order is atomic-arbitrary. */
int oracle_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                             const int *idx, float *grad_points) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *g = grad_out + ((size_t)bi * c + ci) * npoints * nsample;
            const int *ix = idx + (size_t)bi * npoints * nsample;
            float *gp = grad_points + ((size_t)bi * c + ci) * n;
            for (size_t t = 0; t < (size_t)npoints * nsample; ++t) gp[ix[t]] += g[t];
        }
    return 0;
}

/* interpolate_gpu.cu:16-59 — three smallest squared distances over known points, ascending k,
 * strict '<' cascade (:44-55).  bests are double initialised to 1e40 (:37), candidates float (:43);
 * outputs narrowed to float (:57).  The caller applies sqrt (pointnet2_utils.py:98). */
int oracle_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                    int *idx) {
    const int mode = g_dist_mode;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int j = 0; j < n; ++j) {
            const float *u = unknown + ((size_t)bi * n + j) * 3;
            const float *kn = known + (size_t)bi * m * 3;
            const float ux = u[0], uy = u[1], uz = u[2];
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                const float d = sqdist(ux - kn[k * 3 + 0], uy - kn[k * 3 + 1], uz - kn[k * 3 + 2], mode);
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            float *od = dist2 + ((size_t)bi * n + j) * 3;
            int *oi = idx + ((size_t)bi * n + j) * 3;
            od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
            oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
        }
    return 0;
}

/* interpolate_gpu.cu:84-104 — out[b,c,j] = w0*p[i0] + w1*p[i1] + w2*p[i2], left to right.
 * The product/sum rounding is contraction-dependent in the reference; pinned here (and in the
 * HIP kernel) as fma(w2,p2, fma(w1,p1, rn(w0*p0))).  Compared at 1e-4 anyway (features). */
int oracle_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                             const float *weight, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *p = points + ((size_t)bi * c + ci) * m;
            const int *ix = idx + (size_t)bi * n * 3;
            const float *w = weight + (size_t)bi * n * 3;
            float *o = out + ((size_t)bi * c + ci) * n;
            for (int j = 0; j < n; ++j) {
                float t = w[j * 3 + 0] * p[ix[j * 3 + 0]];
                t = fmaf(w[j * 3 + 1], p[ix[j * 3 + 1]], t);
                o[j] = fmaf(w[j * 3 + 2], p[ix[j * 3 + 2]], t);
            }
        }
    return 0;
}

/* interpolate_gpu.cu:127-149 — three adds of grad_out*w_k into caller-zeroed (B,C,M)
 * (pointnet2_utils.py:146).  Sequential order here. */
int oracle_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                  const int *idx, const float *weight, float *grad_points) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *g = grad_out + ((size_t)bi * c + ci) * n;
            const int *ix = idx + (size_t)bi * n * 3;
            const float *w = weight + (size_t)bi * n * 3;
            float *gp = grad_points + ((size_t)bi * c + ci) * m;
            for (int j = 0; j < n; ++j) {
                gp[ix[j * 3 + 0]] += g[j] * w[j * 3 + 0];
                gp[ix[j * 3 + 1]] += g[j] * w[j * 3 + 1];
                gp[ix[j * 3 + 2]] += g[j] * w[j * 3 + 2];
            }
        }
    return 0;
}

/* pointnet2_utils.py:241-264 (QueryAndGroup.forward) as one call:
 * idx = ball_query; out[:, 0:3] = group(xyz^T, idx) - new_xyz^T[..., None]; out[:, 3:] = group(features, idx).
 * out is (B, 3+C, M, ns) with xyz channels first (:257).  features may be NULL (c == 0). */
int oracle_query_and_group(int b, int n, int m, int c, float radius, int nsample,
                           const float *xyz, const float *new_xyz, const float *features,
                           int *idx, float *out) {
    memset(idx, 0, sizeof(int) * (size_t)b * m * nsample); /* caller zero-fill, :218 */
    oracle_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx);
    const size_t ms = (size_t)m * nsample;
#pragma omp parallel for schedule(static)
    for (int bi = 0; bi < b; ++bi) {
        const float *p = xyz + (size_t)bi * n * 3;
        const float *q = new_xyz + (size_t)bi * m * 3;
        const int *ix = idx + (size_t)bi * ms;
        float *o = out + (size_t)bi * (3 + c) * ms;
        for (int a = 0; a < 3; ++a)
            for (int j = 0; j < m; ++j)
                for (int s = 0; s < nsample; ++s)
                    o[a * ms + (size_t)j * nsample + s] = p[ix[(size_t)j * nsample + s] * 3 + a] - q[j * 3 + a];
        for (int ci = 0; ci < c; ++ci) {
            const float *f = features + ((size_t)bi * c + ci) * n;
            for (size_t t = 0; t < ms; ++t) o[(3 + ci) * ms + t] = f[ix[t]];
        }
    }
    return 0;
}
