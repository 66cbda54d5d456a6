/*
 * oracle/pointnet2_stack_oracle.c — CPU restatement of the reference's pointnet2_stack operators
 * (ragged "stacked" batches: points of all samples concatenated, per-sample counts in *_batch_cnt).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as pointnet2_oracle.c): nothing under pdm_ssd_amd/ may use it.
 * Parity status: the reference holds no tests or vectors for these operators ("parity unpinned" by reference
 * fixtures); pinned by hand-derived known answers and by equality with the batch oracle on equal-count batches
 * (tests/test_oracle_kat.py), whose kernels share the arithmetic.
 *
 * Citations are relative to /root/reference/pcdet/ops/pointnet2/pointnet2_stack/src/.
 * Squared distances use the PINNED sequence of the batch oracle, d = fma(dz,dz, fma(dy,dy, rn(dx*dx))).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float sq3(float dx, float dy, float dz) {
    float t = dx * dx;
    t = fmaf(dy, dy, t);
    return fmaf(dz, dz, t);
}

/* sample of flat element `i` given per-sample counts (the kernels' linear scan, e.g. ball_query_gpu.cu:27-32):
 * the LAST sample absorbs everything past the total. */
static int sample_of(int i, int B, const int *cnt, int *start) {
    int bs = 0, acc = cnt[0], s = 0;
    for (int k = 1; k < B; ++k) {
        if (i < acc) break;
        s = acc;
        acc += cnt[k];
        bs = k;
    }
    *start = s;
    return bs;
}

static int prefix(int bs, const int *cnt) {
    int s = 0;
    for (int k = 0; k < bs; ++k) s += cnt[k];
    return s;
}

/* ball_query_gpu.cu:16-66 — idx (M, nsample) LOCAL to the centre's sample, first-nsample-by-index within
 * radius (strict <), padded with the first hit; an empty ball leaves idx[0] = -1 (the python glue then zeroes the
 * row and returns the mask, pointnet2_utils.py:36-37).  idx is caller-zeroed (:33). */
int oracle_stack_ball_query(int B, int M, float radius, int nsample, const float *new_xyz,
                            const int *new_xyz_batch_cnt, const float *xyz, const int *xyz_batch_cnt, int *idx) {
    const float r2 = radius * radius; /* :43 */
#pragma omp parallel for schedule(static)
    for (int pt = 0; pt < M; ++pt) {
        int dummy;
        const int bs = sample_of(pt, B, new_xyz_batch_cnt, &dummy);
        const float *src = xyz + (size_t)prefix(bs, xyz_batch_cnt) * 3;
        const int n = xyz_batch_cnt[bs];
        const float nx = new_xyz[pt * 3], ny = new_xyz[pt * 3 + 1], nz = new_xyz[pt * 3 + 2];
        int *out = idx + (size_t)pt * nsample;
        int cnt = 0;
        for (int k = 0; k < n; ++k) {
            const float d2 = sq3(nx - src[k * 3], ny - src[k * 3 + 1], nz - src[k * 3 + 2]); /* :54 */
            if (d2 < r2) {
                if (cnt == 0)
                    for (int l = 0; l < nsample; ++l) out[l] = k; /* :56-60 */
                out[cnt] = k;
                if (++cnt >= nsample) break;
            }
        }
        if (cnt == 0) out[0] = -1; /* :66 */
    }
    return 0;
}

/* group_points_gpu.cu:67-104 — out (M, C, nsample) = features[start(sample) + idx][c] */
int oracle_stack_group_points(int B, int M, int C, int nsample, const float *features,
                              const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt, float *out) {
#pragma omp parallel for schedule(static)
    for (int pt = 0; pt < M; ++pt) {
        int dummy;
        const int bs = sample_of(pt, B, idx_batch_cnt, &dummy);
        const float *f = features + (size_t)prefix(bs, features_batch_cnt) * C;
        for (int c = 0; c < C; ++c)
            for (int s = 0; s < nsample; ++s)
                out[((size_t)pt * C + c) * nsample + s] = f[(size_t)idx[(size_t)pt * nsample + s] * C + c];
    }
    return 0;
}

/* group_points_gpu.cu:14-44 — grad_features (N, C) += grad_out (M, C, nsample); caller zero-fills */
int oracle_stack_group_points_grad(int B, int M, int C, int N, int nsample, const float *grad_out, const int *idx,
                                   const int *idx_batch_cnt, const int *features_batch_cnt, float *grad_features) {
    (void)N;
    for (int pt = 0; pt < M; ++pt) {
        int dummy;
        const int bs = sample_of(pt, B, idx_batch_cnt, &dummy);
        float *g = grad_features + (size_t)prefix(bs, features_batch_cnt) * C;
        for (int c = 0; c < C; ++c)
            for (int s = 0; s < nsample; ++s)
                g[(size_t)idx[(size_t)pt * nsample + s] * C + c] += grad_out[((size_t)pt * C + c) * nsample + s];
    }
    return 0;
}

/* interpolate_gpu.cu:14-73 — three nearest known points of the same sample; strict-< insertion in index order;
 * best distances start at 1e40 (double) -> +inf after the float store when fewer than three candidates exist;
 * idx is GLOBAL (start of the sample added, :70-72), 0 + start for unfilled slots. */
int oracle_stack_three_nn(int B, int N, const float *unknown, const int *unknown_batch_cnt, const float *known,
                          const int *known_batch_cnt, float *dist2, int *idx) {
#pragma omp parallel for schedule(static)
    for (int pt = 0; pt < N; ++pt) {
        int dummy;
        const int bs = sample_of(pt, B, unknown_batch_cnt, &dummy);
        const int start = prefix(bs, known_batch_cnt), m = known_batch_cnt[bs];
        const float *kn = known + (size_t)start * 3;
        const float ux = unknown[pt * 3], uy = unknown[pt * 3 + 1], uz = unknown[pt * 3 + 2];
        double b1 = 1e40, b2 = 1e40, b3 = 1e40;
        int i1 = 0, i2 = 0, i3 = 0;
        for (int k = 0; k < m; ++k) {
            const float d = sq3(ux - kn[k * 3], uy - kn[k * 3 + 1], uz - kn[k * 3 + 2]);
            if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
            else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
            else if (d < b3) { b3 = d; i3 = k; }
        }
        dist2[pt * 3] = (float)b1; dist2[pt * 3 + 1] = (float)b2; dist2[pt * 3 + 2] = (float)b3;
        idx[pt * 3] = i1 + start; idx[pt * 3 + 1] = i2 + start; idx[pt * 3 + 2] = i3 + start;
    }
    return 0;
}

/* interpolate_gpu.cu:106-125 — out (N, C) = w0 f[i0] + w1 f[i1] + w2 f[i2], written a*b + c*d + e*f there;
 * pinned here (as in the batch oracle) to fma(w2,p2, fma(w1,p1, rn(w0*p0))) */
int oracle_stack_three_interpolate(int N, int C, const float *features, const int *idx, const float *weight, float *out) {
#pragma omp parallel for schedule(static)
    for (int pt = 0; pt < N; ++pt)
        for (int c = 0; c < C; ++c) {
            const float p0 = features[(size_t)idx[pt * 3] * C + c], p1 = features[(size_t)idx[pt * 3 + 1] * C + c],
                        p2 = features[(size_t)idx[pt * 3 + 2] * C + c];
            out[(size_t)pt * C + c] = fmaf(weight[pt * 3 + 2], p2, fmaf(weight[pt * 3 + 1], p1, weight[pt * 3] * p0));
        }
    return 0;
}

/* interpolate_gpu.cu:151-172 — grad_features (M, C) += grad_out (N, C) * w; caller zero-fills */
int oracle_stack_three_interpolate_grad(int N, int C, const float *grad_out, const int *idx, const float *weight,
                                        float *grad_features) {
    for (int pt = 0; pt < N; ++pt)
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < 3; ++k)
                grad_features[(size_t)idx[pt * 3 + k] * C + c] += grad_out[(size_t)pt * C + c] * weight[pt * 3 + k];
    return 0;
}

/* sampling_gpu.cu:187-319 — one 1024-thread block per sample WHATEVER its size (launcher :340), literal LDS tree
 * with the second operand winning only on strict >; idxs are GLOBAL (sample start added, :225,316) and packed
 * sample after sample; temp caller-initialised to 1e10. */
int oracle_stack_furthest_point_sampling(int B, const float *dataset, float *temp, const int *xyz_batch_cnt,
                                         int *idxs, const int *num_sampled_points) {
    enum { S = 1024 };
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int bs = 0; bs < B; ++bs) {
        const int start = prefix(bs, xyz_batch_cnt), ostart = prefix(bs, num_sampled_points);
        const int n = xyz_batch_cnt[bs], m = num_sampled_points[bs];
        const float *ds = dataset + (size_t)start * 3;
        float *tp = temp + start;
        int *out = idxs + ostart;
        float dists[S];
        int dists_i[S];
        if (m <= 0) continue;
        int old = 0;
        out[0] = start; /* :225 */
        for (int j = 1; j < m; ++j) {
            const float x1 = ds[old * 3], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
            for (int tid = 0; tid < S; ++tid) {
                int besti = 0;
                float best = -1;
                for (int k = tid; k < n; k += S) {
                    const float d = sq3(ds[k * 3] - x1, ds[k * 3 + 1] - y1, ds[k * 3 + 2] - z1);
                    const float d2 = fminf(d, tp[k]);
                    tp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int half = S / 2; half >= 1; half >>= 1)
                for (int tid = 0; tid < half; ++tid) {
                    const float v1 = dists[tid], v2 = dists[tid + half];
                    const int i1 = dists_i[tid], i2 = dists_i[tid + half];
                    dists[tid] = v2 > v1 ? v2 : v1;
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            old = dists_i[0];
            out[j] = old + start; /* :316 */
        }
    }
    return err;
}
