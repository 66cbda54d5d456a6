/*
 * oracle/vector_pool_oracle.c — CPU restatement of the reference's voxel query and vector-pool operators
 * (pointnet2_stack: PV-RCNN++'s VectorPool aggregation and the voxel-window neighbour query).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as pointnet2_oracle.c): nothing under pdm_ssd_amd/ may use it.
 * Parity status: "parity unpinned" by reference fixtures — the reference holds no tests or vectors for these
 * operators and its CUDA extension cannot be built here; pinned by hand-derived known answers
 * (tests/test_vector_pool.py) and by cross-checks against the ball-query / three-nn oracles where the
 * semantics coincide.
 *
 * Citations are relative to /root/reference/pcdet/ops/pointnet2/pointnet2_stack/src/.
 * Squared distances use the PINNED sequence of the other oracles, d = fma(dz,dz, fma(dy,dy, rn(dx*dx))), which
 * is also what a contracting compiler makes of the reference's "a*a + b*b + c*c".
 *
 * Where the reference hands out slots of a shared output with atomicAdd on a cursor (vector_pool_gpu.cu:194,
 * :322, :349) the order of the slots is a race; every order is a valid result.  The oracle (and the HIP
 * kernels) take the one order that needs no race: centres in index order, i.e. the cursor is an exclusive prefix
 * sum of the per-centre counts.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float sq3(float dx, float dy, float dz) {
    float t = dx * dx;
    t = fmaf(dy, dy, t);
    return fmaf(dz, dz, t);
}

static int sample_of(int i, int B, const int *cnt) { /* the kernels' linear scan, vector_pool_gpu.cu:146-151 */
    int bs = 0, acc = cnt[0];
    for (int k = 1; k < B; ++k) {
        if (i < acc) break;
        acc += cnt[k];
        bs = k;
    }
    return bs;
}

static int prefix(int bs, const int *cnt) {
    int s = 0;
    for (int k = 0; k < bs; ++k) s += cnt[k];
    return s;
}

/* neighbour test shared by the two scanning kernels (vector_pool_gpu.cu:175-188 and :287-300): ball keeps
 * d2 <= r2, cube keeps |l| <= r on every axis */
static int in_range(int neighbor_type, float lx, float ly, float lz, float r, float r2) {
    if (neighbor_type == 1) return !(sq3(lx, ly, lz) > r2);
    return !((fabsf(lx) > r) | (fabsf(ly) > r) | (fabsf(lz) > r));
}

/* voxel_query_gpu.cu:11-91 — per centre, walk the voxel window around its voxel coordinate (z outer, x inner),
 * keep points within the radius (d2 <= r2), first hit fills the row, -1 in slot 0 when nothing was found.
 * idx is caller-zeroed (voxel_query_utils.py:34). */
int oracle_stack_voxel_query(int M, int R1, int R2, int R3, int nsample, float radius, int z_range, int y_range,
                             int x_range, const float *new_xyz, const float *xyz, const int *new_coords,
                             const int *point_indices, int *idx) {
    const float r2 = radius * radius; /* :26 */
#pragma omp parallel for schedule(static)
    for (int pt = 0; pt < M; ++pt) {
        const float nx = new_xyz[pt * 3], ny = new_xyz[pt * 3 + 1], nz = new_xyz[pt * 3 + 2];
        const int *co = new_coords + (size_t)pt * 4;
        int *out = idx + (size_t)pt * nsample;
        int cnt = 0;
        for (int dz = -z_range; dz <= z_range; ++dz) {
            const int z = co[1] + dz;
            if (z < 0 || z >= R1) continue;
            for (int dy = -y_range; dy <= y_range; ++dy) {
                const int y = co[2] + dy;
                if (y < 0 || y >= R2) continue;
                for (int dx = -x_range; dx <= x_range; ++dx) {
                    const int x = co[3] + dx;
                    if (x < 0 || x >= R3) continue;
                    const int nb = point_indices[(((size_t)co[0] * R1 + z) * R2 + y) * R3 + x]; /* :52-56 */
                    if (nb < 0) continue;
                    const float d2 = sq3(xyz[nb * 3] - nx, xyz[nb * 3 + 1] - ny, xyz[nb * 3 + 2] - nz); /* :63 */
                    if (d2 > r2) continue;
                    if (cnt < nsample) { /* :69-77 */
                        if (cnt == 0)
                            for (int l = 0; l < nsample; ++l) out[l] = nb;
                        out[cnt] = nb;
                        ++cnt;
                    }
                }
            }
        }
        if (cnt == 0) out[0] = -1; /* :90 */
    }
    return 0;
}

/* vector_pool_gpu.cu:125-205 — per centre the GLOBAL indices of its first neighbours in index order: at most
 * nsample when nsample > 0, never more than 1000 (the kernel's temp_idxs[1000], :165-195); start_len (M,2) =
 * [offset in the stack, length]; *cumsum grows by the total; writes stop at capacity = avg_length * M (:197-204).
 * Returns the total (the python caller re-runs with a larger stack while total > capacity). */
int oracle_stack_query_local_neighbor_idxs(const float *support_xyz, const int *xyz_batch_cnt, const float *new_xyz,
                                           const int *new_xyz_batch_cnt, int *stack_neighbor_idxs, int *start_len,
                                           int *cumsum, int avg_length_of_neighbor_idxs, float max_neighbour_distance,
                                           int batch_size, int M, int nsample, int neighbor_type) {
    const float r = max_neighbour_distance, r2 = r * r;
    const long long max_thresh = (long long)avg_length_of_neighbor_idxs * M;
    int *temp = (int *)malloc(sizeof(int) * 1000);
    for (int pt = 0; pt < M; ++pt) {
        const int bs = sample_of(pt, batch_size, new_xyz_batch_cnt);
        const int xstart = prefix(bs, xyz_batch_cnt), n = xyz_batch_cnt[bs];
        const float *src = support_xyz + (size_t)xstart * 3;
        const float nx = new_xyz[pt * 3], ny = new_xyz[pt * 3 + 1], nz = new_xyz[pt * 3 + 2];
        int cnt = 0;
        for (int k = 0; k < n; ++k) {
            if (!in_range(neighbor_type, src[k * 3] - nx, src[k * 3 + 1] - ny, src[k * 3 + 2] - nz, r, r2)) continue;
            if (cnt < 1000) temp[cnt] = k; /* :189-194 */
            else break;
            ++cnt;
            if (nsample > 0 && cnt >= nsample) break;
        }
        const int start = *cumsum; /* :194-195, the atomicAdd taken in centre order */
        *cumsum += cnt;
        start_len[pt * 2] = start;
        start_len[pt * 2 + 1] = cnt;
        if (start >= max_thresh) continue;
        if (start + cnt >= max_thresh) cnt = (int)(max_thresh - start);
        for (int k = 0; k < cnt; ++k) stack_neighbor_idxs[start + k] = temp[k] + xstart;
    }
    free(temp);
    return *cumsum;
}

/* vector_pool_gpu.cu:19-87 — for every (centre, local grid cell) the three nearest of the centre's stacked
 * neighbours to the cell centre: strict-< insertion in list order, double bests from 1e40 (+inf after the float
 * store), idx -1 when the list is empty, the best duplicated into unfilled second / third slots. */
int oracle_stack_three_nn_by_local_idxs(const float *support_xyz, const float *new_xyz_grid_centers, int *grid_idxs,
                                        float *grid_dist2, const int *stack_neighbor_idxs, const int *start_len, int M,
                                        int num_total_grids) {
#pragma omp parallel for schedule(static)
    for (int pt = 0; pt < M; ++pt) {
        const int *list = stack_neighbor_idxs + start_len[pt * 2];
        const int len = start_len[pt * 2 + 1];
        for (int g = 0; g < num_total_grids; ++g) {
            const size_t o = ((size_t)pt * num_total_grids + g) * 3;
            const float cx = new_xyz_grid_centers[o], cy = new_xyz_grid_centers[o + 1], cz = new_xyz_grid_centers[o + 2];
            double b1 = 1e40, b2 = 1e40, b3 = 1e40;
            int i1 = -1, i2 = -1, i3 = -1;
            for (int k = 0; k < len; ++k) {
                const int nb = list[k];
                const float d = sq3(cx - support_xyz[nb * 3], cy - support_xyz[nb * 3 + 1], cz - support_xyz[nb * 3 + 2]);
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = nb; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = nb; }
                else if (d < b3) { b3 = d; i3 = nb; }
            }
            if (i2 == -1) { i2 = i1; b2 = b1; } /* :73-78 */
            if (i3 == -1) { i3 = i1; b3 = b1; }
            grid_dist2[o] = (float)b1; grid_dist2[o + 1] = (float)b2; grid_dist2[o + 2] = (float)b3;
            grid_idxs[o] = i1; grid_idxs[o + 1] = i2; grid_idxs[o + 2] = i3;
        }
    }
    return 0;
}

/* vector_pool_gpu.cu:245-361 + the launcher :364-413 — per centre, neighbours in index order fall into the cells
 * of its local num_grid_x * num_grid_y * num_grid_z lattice; pooling 0 sums features (input channel i folds onto
 * i % num_c_each_grid) and local offsets per cell, pooling 1 keeps the first point of every cell.  Outputs are
 * caller-zeroed SUMS (python divides by the counts, pointnet2_utils.py:416-420).  grouped_idxs (cap,3) =
 * [support idx, centre, cell] for the backward; returns the number of entries wanted (python re-runs with a larger
 * cap when it exceeds num_max_sum_points). */
int oracle_stack_vector_pool(const float *support_xyz, const float *support_features, const int *xyz_batch_cnt,
                             const float *new_xyz, float *new_features, float *new_local_xyz,
                             const int *new_xyz_batch_cnt, int *point_cnt_of_grid, int *grouped_idxs, int num_grid_x,
                             int num_grid_y, int num_grid_z, float max_neighbour_distance, int batch_size, int M,
                             int num_c_in, int num_c_out, int use_xyz, int num_max_sum_points, int nsample,
                             int neighbor_type, int pooling_type) {
    const int G = num_grid_x * num_grid_y * num_grid_z;
    const int ceg = num_c_out / G; /* :384 */
    const float r = max_neighbour_distance, r2 = r * r;
    const float gsx = r * 2 / num_grid_x, gsy = r * 2 / num_grid_y, gsz = r * 2 / num_grid_z; /* :385-387 */
    int cum = 0;
    for (int pt = 0; pt < M; ++pt) {
        const int bs = sample_of(pt, batch_size, new_xyz_batch_cnt);
        const int xstart = prefix(bs, xyz_batch_cnt), n = xyz_batch_cnt[bs];
        const float *src = support_xyz + (size_t)xstart * 3;
        const float *feat = support_features + (size_t)xstart * num_c_in;
        const float nx = new_xyz[pt * 3], ny = new_xyz[pt * 3 + 1], nz = new_xyz[pt * 3 + 2];
        float *nf = new_features + (size_t)pt * num_c_out;
        float *nl = new_local_xyz + (size_t)pt * 3 * G;
        int *pc = point_cnt_of_grid + (size_t)pt * G;
        int sample_cnt = 0;
        for (int k = 0; k < n; ++k) {
            const float lx = src[k * 3] - nx, ly = src[k * 3 + 1] - ny, lz = src[k * 3 + 2] - nz;
            if (!in_range(neighbor_type, lx, ly, lz, r, r2)) continue;
            const int gx = (int)floorf((lx + r) / gsx), gy = (int)floorf((ly + r) / gsy), gz = (int)floorf((lz + r) / gsz);
            int g = gx * num_grid_y * num_grid_z + gy * num_grid_z + gz; /* :305 */
            g = g < 0 ? 0 : g > G - 1 ? G - 1 : g;
            if (pooling_type == 0) {
                pc[g]++;
                for (int i = 0; i < num_c_in; ++i) nf[g * ceg + i % ceg] += feat[(size_t)k * num_c_in + i];
                if (use_xyz) { nl[g * 3] += lx; nl[g * 3 + 1] += ly; nl[g * 3 + 2] += lz; }
            } else {
                if (pc[g] != 0) continue; /* :332 */
                pc[g]++;
                for (int i = 0; i < num_c_in; ++i) nf[g * ceg + i % ceg] = feat[(size_t)k * num_c_in + i];
                if (use_xyz) { nl[g * 3] = lx; nl[g * 3 + 1] = ly; nl[g * 3 + 2] = lz; }
            }
            const int cnt = cum++; /* :322 / :349 */
            if (cnt >= num_max_sum_points) continue; /* keeps counting; the caller retries */
            grouped_idxs[cnt * 3] = xstart + k;
            grouped_idxs[cnt * 3 + 1] = pt;
            grouped_idxs[cnt * 3 + 2] = g;
            ++sample_cnt;
            if (nsample > 0 && sample_cnt >= nsample) break;
            if (pooling_type == 1 && sample_cnt >= G) break; /* :356 */
        }
    }
    return cum;
}

/* vector_pool_gpu.cu:416-443 — grad_support_features (N, C_in), caller-zeroed: every grouped entry adds
 * grad_new[centre][cell * ceg + c % ceg] * (1 / max(count, 1)) to channel c of its support point.  The reference
 * adds with float atomics (order undefined); the oracle accumulates in double and rounds once. */
int oracle_stack_vector_pool_grad(const float *grad_new_features, const int *point_cnt_of_grid, const int *grouped_idxs,
                                  float *grad_support_features, int N, int M, int num_c_out, int num_c_in,
                                  int num_total_grids, int num_entries) {
    (void)M;
    const int ceg = num_c_out / num_total_grids;
    double *acc = (double *)calloc((size_t)N * num_c_in, sizeof(double));
    for (int e = 0; e < num_entries; ++e) {
        const int k = grouped_idxs[e * 3], pt = grouped_idxs[e * 3 + 1], g = grouped_idxs[e * 3 + 2];
        const float w = 1 / fmaxf((float)point_cnt_of_grid[(size_t)pt * num_total_grids + g], 1.0f); /* :441 */
        const float *gn = grad_new_features + (size_t)pt * num_c_out + (size_t)g * ceg;
        for (int c = 0; c < num_c_in; ++c) acc[(size_t)k * num_c_in + c] += (double)(gn[c % ceg] * w);
    }
    for (size_t i = 0; i < (size_t)N * num_c_in; ++i) grad_support_features[i] = (float)acc[i];
    free(acc);
    return 0;
}
