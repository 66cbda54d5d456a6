// Voxel-window neighbour query and the vector-pool family of pointnet2_stack (reference:
// pcdet/ops/pointnet2/pointnet2_stack/src/voxel_query_gpu.cu, vector_pool_gpu.cu).  SURVEY.md section 8(f) row N3.
//
// The reference runs ONE THREAD per centre over all points of its sample and hands out slots of the shared outputs
// (stacked neighbour lists, grouped_idxs) with atomicAdd on a global cursor, re-running the whole kernel from python
// with a larger buffer when the cursor overran it.  Here a WAVE owns a centre — 64 candidates per step, ballot-ordered
// so "first by index" is kept — and the cursor is an exclusive prefix sum of per-centre counts (count -> scan ->
// fill), which is one of the orders the reference's race allows, is the same on every run, and lets the python side
// size the buffers exactly instead of retrying.
#include "common.h"

namespace pdm {

// vector_pool_gpu.cu:175-188 / :287-300 — ball keeps d2 <= r2, cube keeps |l| <= r on every axis
__device__ __forceinline__ bool vp_in_range(int neighbor_type, float lx, float ly, float lz, float r, float r2) {
    if (neighbor_type == 1) return !(sqdist(lx, ly, lz) > r2);
    return !(fabsf(lx) > r || fabsf(ly) > r || fabsf(lz) > r);
}

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// ---- voxel query (voxel_query_gpu.cu:11-91): the window cells in the reference's order (z outer, x inner), 64 per
// step; cells outside the volume or empty are simply not hits -------------------------------------------------------
__global__ __launch_bounds__(256) void voxel_query_kernel(int M, int R1, int R2, int R3, int nsample, float radius, int z_range,
                                                          int y_range, int x_range, const float *__restrict__ new_xyz,
                                                          const float *__restrict__ xyz, const int *__restrict__ new_coords,
                                                          const int *__restrict__ point_indices, int *__restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pt >= M) return;
    const float r2 = __fmul_rn(radius, radius);
    const float nx = new_xyz[(size_t)pt * 3], ny = new_xyz[(size_t)pt * 3 + 1], nz = new_xyz[(size_t)pt * 3 + 2];
    const int b = new_coords[(size_t)pt * 4], cz = new_coords[(size_t)pt * 4 + 1], cy = new_coords[(size_t)pt * 4 + 2],
              cx = new_coords[(size_t)pt * 4 + 3];
    const int wy = 2 * y_range + 1, wx = 2 * x_range + 1;
    const int window = (2 * z_range + 1) * wy * wx;
    int *__restrict__ out = idx + (size_t)pt * nsample;
    int cnt = 0, first = -1;
    for (int base = 0; base < window && cnt < nsample; base += 64) {
        const int w = base + lane;
        int nb = -1;
        if (w < window) {
            const int z = cz + w / (wy * wx) - z_range, y = cy + (w / wx) % wy - y_range, x = cx + w % wx - x_range;
            if (z >= 0 && z < R1 && y >= 0 && y < R2 && x >= 0 && x < R3)
                nb = point_indices[(((size_t)b * R1 + z) * R2 + y) * R3 + x];
        }
        bool hit = false;
        if (nb >= 0)
            hit = !(sqdist(xyz[(size_t)nb * 3] - nx, xyz[(size_t)nb * 3 + 1] - ny, xyz[(size_t)nb * 3 + 2] - nz) > r2);
        const unsigned long long mask = __ballot(hit);
        if (mask == 0) continue;
        if (cnt == 0) first = __shfl(nb, __ffsll((long long)mask) - 1, 64);
        const int rank = cnt + __popcll(mask & lanes_below(lane));
        if (hit && rank < nsample) out[rank] = nb;
        cnt += __popcll(mask);
    }
    if (cnt == 0) {
        if (lane == 0) out[0] = -1;   // :90; the python glue zeroes the row and returns the mask
    } else {
        for (int l = cnt + lane; l < nsample; l += 64) out[l] = first;   // :70-74 the first hit fills the row
    }
}

// ---- stacked local neighbour lists (vector_pool_gpu.cu:125-205) ---------------------------------------------------
// FILL = false: start_len[pt][1] = number of neighbours kept (first by index; at most nsample when nsample > 0, never
// more than 1000 — the reference's temp_idxs[1000]).  FILL = true: write them at start_len[pt][0], stopping at the
// capacity of the stack (:197-204).
template <bool FILL>
__global__ __launch_bounds__(256) void local_neighbors_kernel(int B, int M, float r, int nsample, int neighbor_type,
                                                              const float *__restrict__ support_xyz,
                                                              const int *__restrict__ xyz_cnt,
                                                              const float *__restrict__ new_xyz,
                                                              const int *__restrict__ new_cnt, int *start_len,
                                                              int *__restrict__ stack, long long max_thresh) {
    __shared__ int s_new[ST_MAXB + 1], s_xyz[ST_MAXB + 1];
    build_prefix_pair(B, new_cnt, s_new, xyz_cnt, s_xyz);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pt >= M) return;
    const int bs = sample_of(pt, B, s_new);
    const int xstart = s_xyz[bs], n = s_xyz[bs + 1] - s_xyz[bs];
    const float *__restrict__ src = support_xyz + (size_t)xstart * 3;
    const float r2 = __fmul_rn(r, r);
    const float nx = new_xyz[(size_t)pt * 3], ny = new_xyz[(size_t)pt * 3 + 1], nz = new_xyz[(size_t)pt * 3 + 2];
    int cap = nsample > 0 && nsample < 1000 ? nsample : 1000;
    long long start = 0;
    if (FILL) {
        start = start_len[(size_t)pt * 2];
        const int len = start_len[(size_t)pt * 2 + 1];
        if (start >= max_thresh) return;
        cap = start + len >= max_thresh ? (int)(max_thresh - start) : len;
    }
    int cnt = 0;
    for (int base = 0; base < n && cnt < cap; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n)
            hit = vp_in_range(neighbor_type, src[(size_t)k * 3] - nx, src[(size_t)k * 3 + 1] - ny, src[(size_t)k * 3 + 2] - nz, r, r2);
        const unsigned long long mask = __ballot(hit);
        if (FILL) {
            const int rank = cnt + __popcll(mask & lanes_below(lane));
            if (hit && rank < cap) stack[start + rank] = k + xstart;
        }
        cnt += __popcll(mask);
    }
    if (!FILL && lane == 0) start_len[(size_t)pt * 2 + 1] = cnt < cap ? cnt : cap;
}

// start[i * ss] = *cumsum + sum of cnt[j * cs] for j < i; *cumsum += the total.  One workgroup (M is a number of
// centres).  cnt and start may be the two columns of one (M, 2) array.
__global__ __launch_bounds__(1024) void cursor_scan_kernel(const int *cnt, int cs, int *start, int ss, int M, int *cumsum) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = cumsum[0];
    __syncthreads();
    for (int base = 0; base < M; base += 1024) {
        const int i = base + tid;
        const int v = i < M ? cnt[(size_t)i * cs] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < M) start[(size_t)i * ss] = before + incl - v;
        __syncthreads();
        if (tid == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) cumsum[0] = s_carry;
}

// ---- three nearest stacked neighbours of every local grid-cell centre (vector_pool_gpu.cu:19-87): lanes over the
// cells of one centre, the list read wave-uniformly -------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_nn_local_kernel(int M, int G, const float *__restrict__ support_xyz,
                                                             const float *__restrict__ centers, int *__restrict__ idxs,
                                                             float *__restrict__ dist2, const int *__restrict__ stack,
                                                             const int *__restrict__ start_len, long long stack_len) {
    const int lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pt >= M) return;
    const long long start = start_len[(size_t)pt * 2];
    int len = start_len[(size_t)pt * 2 + 1];
    if (start + len > stack_len) len = start < stack_len ? (int)(stack_len - start) : 0;   // an overrun stack (caller retries)
    const int *__restrict__ list = stack + start;
    const float INF = __builtin_inff();   // (float)1e40 — see stack_three_nn_kernel
    for (int g = lane; g < G; g += 64) {
        const size_t o = ((size_t)pt * G + g) * 3;
        const float cx = centers[o], cy = centers[o + 1], cz = centers[o + 2];
        float b1 = INF, b2 = INF, b3 = INF;
        int i1 = -1, i2 = -1, i3 = -1;
        for (int k = 0; k < len; ++k) {
            const int nb = list[k];
            const float d = sqdist(cx - support_xyz[(size_t)nb * 3], cy - support_xyz[(size_t)nb * 3 + 1],
                                   cz - support_xyz[(size_t)nb * 3 + 2]);
            if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = nb; }
            else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = nb; }
            else if (d < b3) { b3 = d; i3 = nb; }
        }
        if (i2 == -1) { i2 = i1; b2 = b1; }   // :73-78
        if (i3 == -1) { i3 = i1; b3 = b1; }
        dist2[o] = b1; dist2[o + 1] = b2; dist2[o + 2] = b3;
        idxs[o] = i1; idxs[o + 1] = i2; idxs[o + 2] = i3;
    }
}

// ---- vector pool (vector_pool_gpu.cu:245-361) --------------------------------------------------------------------
struct VpArgs {
    const float *support_xyz, *support_features, *new_xyz;
    const int *xyz_cnt, *new_cnt;
    float *new_features, *new_local_xyz;
    int *point_cnt_of_grid, *grouped_idxs;
    int *entry_cnt;           // COUNT: (M) entries each centre records
    const int *entry_start;   // POOL: (M) exclusive prefix of entry_cnt
    int B, M, gx, gy, gz, G, c_in, c_out, ceg, use_xyz, max_entries, nsample, neighbor_type, pooling_type;
    float r, gsx, gsy, gsz;
    int waves;   // centres per workgroup
};

// One wave per centre; the wave's slice of dynamic LDS holds the centre's output row while it is being summed:
// acc[c_out] | lxyz[3G] | pcnt[G].  Candidates are tested 64 at a time; the kept ones are then folded one after the
// other in index order (lanes over the output channels of the cell), which is the order — hence the rounding — of
// the reference's per-thread loop.
template <bool COUNT>
__global__ __launch_bounds__(256) void vector_pool_kernel(VpArgs a) {
    __shared__ int s_new[ST_MAXB + 1], s_xyz[ST_MAXB + 1];
    extern __shared__ float vp_lds[];
    build_prefix_pair(a.B, a.new_cnt, s_new, a.xyz_cnt, s_xyz);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = blockIdx.x * a.waves + wave;
    if (wave >= a.waves || pt >= a.M) return;
    const int G = a.G, ceg = a.ceg, c_in = a.c_in;
    float *acc = vp_lds + (size_t)wave * (a.c_out + 4 * G);
    float *lxyz = acc + a.c_out;
    int *pcnt = reinterpret_cast<int *>(lxyz + 3 * G);
    for (int j = lane; j < a.c_out + 4 * G; j += 64) acc[j] = 0.f;   // int 0 == float 0 bits
    __builtin_amdgcn_wave_barrier();

    const int bs = sample_of(pt, a.B, s_new);
    const int xstart = s_xyz[bs], n = s_xyz[bs + 1] - s_xyz[bs];
    const float *__restrict__ src = a.support_xyz + (size_t)xstart * 3;
    const float *__restrict__ feat = a.support_features + (size_t)xstart * c_in;
    const float r = a.r, r2 = __fmul_rn(r, r);
    const float nx = a.new_xyz[(size_t)pt * 3], ny = a.new_xyz[(size_t)pt * 3 + 1], nz = a.new_xyz[(size_t)pt * 3 + 2];
    int remaining = a.nsample > 0 ? a.nsample : 0x7fffffff;
    if (a.pooling_type == 1 && G < remaining) remaining = G;   // :356
    const long long estart = COUNT ? 0 : a.entry_start[pt];
    int taken = 0;

    for (int base = 0; base < n && taken < remaining; base += 64) {
        const int k = base + lane;
        bool hit = false;
        float lx = 0.f, ly = 0.f, lz = 0.f;
        int g = 0;
        if (k < n) {
            lx = src[(size_t)k * 3] - nx; ly = src[(size_t)k * 3 + 1] - ny; lz = src[(size_t)k * 3 + 2] - nz;
            hit = vp_in_range(a.neighbor_type, lx, ly, lz, r, r2);
            // :302-306 — true division, floor, linear index clamped into the lattice
            const int ix = (int)floorf(__fdiv_rn(lx + r, a.gsx)), iy = (int)floorf(__fdiv_rn(ly + r, a.gsy)),
                      iz = (int)floorf(__fdiv_rn(lz + r, a.gsz));
            g = ix * a.gy * a.gz + iy * a.gz + iz;
            g = g < 0 ? 0 : g > G - 1 ? G - 1 : g;
        }
        unsigned long long mask = __ballot(hit);
        if (mask == 0) continue;
        if (a.pooling_type == 0) {
            const int rank = __popcll(mask & lanes_below(lane));
            const bool take = hit && taken + rank < remaining;
            mask = __ballot(take);
            if (COUNT) { taken += __popcll(mask); continue; }
            const long long e = estart + taken + rank;
            if (take && e < a.max_entries) {
                a.grouped_idxs[e * 3] = xstart + k; a.grouped_idxs[e * 3 + 1] = pt; a.grouped_idxs[e * 3 + 2] = g;
            }
            taken += __popcll(mask);
            while (mask) {
                const int bsel = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int kb = __shfl(k, bsel, 64), gb = __shfl(g, bsel, 64);
                const float sx = __shfl(lx, bsel, 64), sy = __shfl(ly, bsel, 64), sz = __shfl(lz, bsel, 64);   // all lanes take part
                const float l = lane == 0 ? sx : lane == 1 ? sy : sz;
                if (lane == 0) pcnt[gb]++;
                if (a.use_xyz && lane < 3) lxyz[gb * 3 + lane] += l;
                const float *__restrict__ row = feat + (size_t)kb * c_in;
                for (int rr = lane; rr < ceg; rr += 64) {
                    float s = acc[gb * ceg + rr];
                    for (int i = rr; i < c_in; i += ceg) s += row[i];   // channel i folds onto i % ceg, ascending i
                    acc[gb * ceg + rr] = s;
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            while (mask && taken < remaining) {
                const int bsel = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int gb = __shfl(g, bsel, 64);
                if (pcnt[gb] != 0) continue;   // :332 only the first point of a cell
                __builtin_amdgcn_wave_barrier();
                const int kb = __shfl(k, bsel, 64);
                const float sx = __shfl(lx, bsel, 64), sy = __shfl(ly, bsel, 64), sz = __shfl(lz, bsel, 64);   // all lanes take part
                const float l = lane == 0 ? sx : lane == 1 ? sy : sz;
                if (lane == 0) pcnt[gb] = 1;
                if (!COUNT) {
                    if (a.use_xyz && lane < 3) lxyz[gb * 3 + lane] = l;
                    const float *__restrict__ row = feat + (size_t)kb * c_in;
                    for (int rr = lane; rr < ceg; rr += 64) {   // '=' per channel: the last i with i % ceg == rr stays
                        const int last = rr + ((c_in - 1 - rr) / ceg) * ceg;
                        if (rr < c_in) acc[gb * ceg + rr] = row[last];
                    }
                    const long long e = estart + taken;
                    if (lane == 0 && e < a.max_entries) {
                        a.grouped_idxs[e * 3] = xstart + kb; a.grouped_idxs[e * 3 + 1] = pt; a.grouped_idxs[e * 3 + 2] = gb;
                    }
                }
                ++taken;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (COUNT) {
        if (lane == 0) a.entry_cnt[pt] = taken;
        return;
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < a.c_out; j += 64) a.new_features[(size_t)pt * a.c_out + j] = acc[j];
    for (int j = lane; j < 3 * G; j += 64) a.new_local_xyz[(size_t)pt * 3 * G + j] = lxyz[j];
    for (int j = lane; j < G; j += 64) a.point_cnt_of_grid[(size_t)pt * G + j] = pcnt[j];
}

// vector_pool_gpu.cu:416-443 — one thread per (entry, input channel), channels fastest
__global__ __launch_bounds__(256) void vector_pool_grad_kernel(long long total, int c_in, int c_out, int ceg, int G,
                                                               const float *__restrict__ grad_new,
                                                               const int *__restrict__ point_cnt,
                                                               const int *__restrict__ grouped,
                                                               float *__restrict__ grad_support) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long e = t / c_in;
        const int c = (int)(t - e * c_in);
        const int k = grouped[e * 3], pt = grouped[e * 3 + 1], g = grouped[e * 3 + 2];
        const float w = __fdiv_rn(1.0f, fmaxf((float)point_cnt[(size_t)pt * G + g], 1.0f));
        atomicAdd(grad_support + (size_t)k * c_in + c, __fmul_rn(grad_new[(size_t)pt * c_out + (size_t)g * ceg + c % ceg], w));
    }
}

static int vp_fill_args(VpArgs &a, const char *who, int num_grid_x, int num_grid_y, int num_grid_z, float max_neighbour_distance,
                        int batch_size, int M, int num_c_in, int num_c_out, size_t *lds) {
    PDM_REQUIRE(batch_size >= 1 && batch_size <= ST_MAXB && M >= 0, PDM_E_BADARG, "%s: batch_size=%d M=%d", who, batch_size, M);
    PDM_REQUIRE(num_grid_x > 0 && num_grid_y > 0 && num_grid_z > 0 && (long long)num_grid_x * num_grid_y * num_grid_z <= 4096,
                PDM_E_BADARG, "%s: local lattice %dx%dx%d", who, num_grid_x, num_grid_y, num_grid_z);
    a.G = num_grid_x * num_grid_y * num_grid_z;
    PDM_REQUIRE(num_c_in > 0 && num_c_out > 0 && num_c_out % a.G == 0, PDM_E_BADARG,
                "%s: num_c_out=%d is not a multiple of the %d cells", who, num_c_out, a.G);
    a.B = batch_size; a.M = M; a.gx = num_grid_x; a.gy = num_grid_y; a.gz = num_grid_z;
    a.c_in = num_c_in; a.c_out = num_c_out; a.ceg = num_c_out / a.G;
    a.r = max_neighbour_distance;
    a.gsx = max_neighbour_distance * 2 / num_grid_x;   // vector_pool_gpu.cu:385-387
    a.gsy = max_neighbour_distance * 2 / num_grid_y;
    a.gsz = max_neighbour_distance * 2 / num_grid_z;
    const size_t per_wave = ((size_t)num_c_out + 4 * (size_t)a.G) * sizeof(float);
    PDM_REQUIRE(per_wave <= 48 * 1024, PDM_E_TOOLARGE, "%s: an output row of %d channels + %d cells does not fit one wave's LDS",
                who, num_c_out, a.G);
    a.waves = per_wave * 4 <= 48 * 1024 ? 4 : per_wave * 2 <= 48 * 1024 ? 2 : 1;
    *lds = per_wave * a.waves;
    return 0;
}

}  // namespace pdm

using namespace pdm;

extern "C" int pdm_stack_voxel_query(void *stream, int M, int Z, int Y, int X, int nsample, float radius, int z_range,
                                     int y_range, int x_range, const float *new_xyz, const float *xyz, const int *new_coords,
                                     const int *point_indices, int *idx) {
    PDM_REQUIRE(M >= 0 && Z > 0 && Y > 0 && X > 0 && nsample > 0 && z_range >= 0 && y_range >= 0 && x_range >= 0, PDM_E_BADARG,
                "stack_voxel_query: M=%d volume %dx%dx%d nsample=%d range (%d,%d,%d)", M, Z, Y, X, nsample, z_range, y_range, x_range);
    PDM_REQUIRE((long long)(2 * z_range + 1) * (2 * y_range + 1) * (2 * x_range + 1) < (1ll << 30), PDM_E_BADARG,
                "stack_voxel_query: window too large");
    if (M == 0) return 0;
    PDM_REQUIRE(new_xyz && xyz && new_coords && point_indices && idx, PDM_E_BADARG, "stack_voxel_query: null pointer");
    hipLaunchKernelGGL(voxel_query_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), M, Z, Y, X, nsample, radius, z_range,
                       y_range, x_range, new_xyz, xyz, new_coords, point_indices, idx);
    return check_launch("stack_voxel_query");
}

static int local_neighbors_common(const char *who, int batch_size, int M, const void *p0, const void *p1, const void *p2,
                                  const void *p3, const void *p4) {
    PDM_REQUIRE(batch_size >= 1 && batch_size <= ST_MAXB && M >= 0, PDM_E_BADARG, "%s: batch_size=%d M=%d", who, batch_size, M);
    PDM_REQUIRE(M == 0 || (p0 && p1 && p2 && p3 && p4), PDM_E_BADARG, "%s: null pointer", who);
    return 0;
}

extern "C" int pdm_stack_local_neighbor_count(void *stream, const float *support_xyz, const int *xyz_batch_cnt,
                                              const float *new_xyz, const int *new_xyz_batch_cnt, int *start_len, int *cumsum,
                                              float max_neighbour_distance, int batch_size, int M, int nsample, int neighbor_type) {
    if (int rc = local_neighbors_common("stack_local_neighbor_count", batch_size, M, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                                        start_len, cumsum))
        return rc;
    if (M == 0) return 0;
    hipLaunchKernelGGL(local_neighbors_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), batch_size, M,
                       max_neighbour_distance, nsample, neighbor_type, support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                       start_len, (int *)nullptr, 0ll);
    hipLaunchKernelGGL(cursor_scan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), start_len + 1, 2, start_len, 2, M, cumsum);
    return check_launch("stack_local_neighbor_count");
}

extern "C" int pdm_stack_local_neighbor_fill(void *stream, const float *support_xyz, const int *xyz_batch_cnt,
                                             const float *new_xyz, const int *new_xyz_batch_cnt, int *stack_neighbor_idxs,
                                             const int *start_len, long long stack_capacity, float max_neighbour_distance,
                                             int batch_size, int M, int nsample, int neighbor_type) {
    if (int rc = local_neighbors_common("stack_local_neighbor_fill", batch_size, M, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                                        start_len, start_len))
        return rc;
    if (M == 0 || stack_capacity <= 0) return 0;
    PDM_REQUIRE(stack_neighbor_idxs, PDM_E_BADARG, "stack_local_neighbor_fill: null stack");
    hipLaunchKernelGGL(local_neighbors_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), batch_size, M,
                       max_neighbour_distance, nsample, neighbor_type, support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                       const_cast<int *>(start_len), stack_neighbor_idxs, stack_capacity);
    return check_launch("stack_local_neighbor_fill");
}

extern "C" int pdm_stack_query_local_neighbor_idxs(void *stream, const float *support_xyz, const int *xyz_batch_cnt,
                                                   const float *new_xyz, const int *new_xyz_batch_cnt, int *stack_neighbor_idxs,
                                                   int *start_len, int *cumsum, int avg_length_of_neighbor_idxs,
                                                   float max_neighbour_distance, int batch_size, int M, int nsample,
                                                   int neighbor_type) {
    PDM_REQUIRE(avg_length_of_neighbor_idxs >= 0, PDM_E_BADARG, "stack_query_local_neighbor_idxs: avg_length=%d", avg_length_of_neighbor_idxs);
    if (int rc = pdm_stack_local_neighbor_count(stream, support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, start_len, cumsum,
                                                max_neighbour_distance, batch_size, M, nsample, neighbor_type))
        return rc;
    return pdm_stack_local_neighbor_fill(stream, support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, stack_neighbor_idxs,
                                         start_len, (long long)avg_length_of_neighbor_idxs * M, max_neighbour_distance, batch_size, M,
                                         nsample, neighbor_type);
}

extern "C" int pdm_stack_three_nn_by_local_idxs(void *stream, const float *support_xyz, const float *new_xyz_grid_centers,
                                                int *new_xyz_grid_idxs, float *new_xyz_grid_dist2, const int *stack_neighbor_idxs,
                                                const int *start_len, long long stack_len, int M, int num_total_grids) {
    PDM_REQUIRE(M >= 0 && num_total_grids >= 0 && stack_len >= 0, PDM_E_BADARG, "stack_three_nn_by_local_idxs: M=%d grids=%d", M, num_total_grids);
    if (M == 0 || num_total_grids == 0) return 0;
    PDM_REQUIRE(new_xyz_grid_centers && new_xyz_grid_idxs && new_xyz_grid_dist2 && start_len && (stack_len == 0 || (support_xyz && stack_neighbor_idxs)),
                PDM_E_BADARG, "stack_three_nn_by_local_idxs: null pointer");
    hipLaunchKernelGGL(three_nn_local_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), M, num_total_grids, support_xyz,
                       new_xyz_grid_centers, new_xyz_grid_idxs, new_xyz_grid_dist2, stack_neighbor_idxs, start_len, stack_len);
    return check_launch("stack_three_nn_by_local_idxs");
}

extern "C" int pdm_stack_vector_pool_count(void *stream, const float *support_xyz, const int *xyz_batch_cnt, const float *new_xyz,
                                           const int *new_xyz_batch_cnt, int *entry_start, int *entry_cnt, int *total,
                                           int num_grid_x, int num_grid_y, int num_grid_z, float max_neighbour_distance,
                                           int batch_size, int M, int nsample, int neighbor_type, int pooling_type) {
    VpArgs a{};
    size_t lds = 0;
    // the count pass needs the cells' "seen" flags only: size its LDS for one channel per cell
    const int G = num_grid_x * num_grid_y * num_grid_z;
    if (int rc = vp_fill_args(a, "stack_vector_pool_count", num_grid_x, num_grid_y, num_grid_z, max_neighbour_distance, batch_size, M,
                              1, G > 0 ? G : 1, &lds))
        return rc;
    PDM_REQUIRE(pooling_type == 0 || pooling_type == 1, PDM_E_BADARG, "stack_vector_pool_count: pooling_type=%d", pooling_type);
    PDM_REQUIRE(total, PDM_E_BADARG, "stack_vector_pool_count: null total");
    if (M == 0) return 0;
    PDM_REQUIRE(support_xyz && xyz_batch_cnt && new_xyz && new_xyz_batch_cnt && entry_start && entry_cnt, PDM_E_BADARG,
                "stack_vector_pool_count: null pointer");
    a.support_xyz = support_xyz; a.support_features = nullptr; a.new_xyz = new_xyz; a.xyz_cnt = xyz_batch_cnt; a.new_cnt = new_xyz_batch_cnt;
    a.entry_cnt = entry_cnt; a.nsample = nsample; a.neighbor_type = neighbor_type; a.pooling_type = pooling_type;
    hipLaunchKernelGGL(vector_pool_kernel<true>, dim3((M + a.waves - 1) / a.waves), dim3(256), lds, as_stream(stream), a);
    hipLaunchKernelGGL(cursor_scan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), entry_cnt, 1, entry_start, 1, M, total);
    return check_launch("stack_vector_pool_count");
}

extern "C" int pdm_stack_vector_pool(void *stream, const float *support_xyz, const float *support_features, const int *xyz_batch_cnt,
                                     const float *new_xyz, float *new_features, float *new_local_xyz, const int *new_xyz_batch_cnt,
                                     int *point_cnt_of_grid, int *grouped_idxs, const int *entry_start, int num_grid_x,
                                     int num_grid_y, int num_grid_z, float max_neighbour_distance, int batch_size, int M,
                                     int num_c_in, int num_c_out, int use_xyz, int num_max_sum_points, int nsample, int neighbor_type,
                                     int pooling_type) {
    VpArgs a{};
    size_t lds = 0;
    if (int rc = vp_fill_args(a, "stack_vector_pool", num_grid_x, num_grid_y, num_grid_z, max_neighbour_distance, batch_size, M,
                              num_c_in, num_c_out, &lds))
        return rc;
    PDM_REQUIRE(pooling_type == 0 || pooling_type == 1, PDM_E_BADARG, "stack_vector_pool: pooling_type=%d", pooling_type);
    PDM_REQUIRE(num_max_sum_points >= 0, PDM_E_BADARG, "stack_vector_pool: num_max_sum_points=%d", num_max_sum_points);
    if (M == 0) return 0;
    PDM_REQUIRE(support_xyz && support_features && xyz_batch_cnt && new_xyz && new_xyz_batch_cnt && new_features && new_local_xyz &&
                    point_cnt_of_grid && entry_start && (grouped_idxs || num_max_sum_points == 0),
                PDM_E_BADARG, "stack_vector_pool: null pointer");
    a.support_xyz = support_xyz; a.support_features = support_features; a.new_xyz = new_xyz; a.xyz_cnt = xyz_batch_cnt;
    a.new_cnt = new_xyz_batch_cnt; a.new_features = new_features; a.new_local_xyz = new_local_xyz;
    a.point_cnt_of_grid = point_cnt_of_grid; a.grouped_idxs = grouped_idxs; a.entry_start = entry_start;
    a.use_xyz = use_xyz; a.max_entries = num_max_sum_points; a.nsample = nsample; a.neighbor_type = neighbor_type;
    a.pooling_type = pooling_type;
    hipLaunchKernelGGL(vector_pool_kernel<false>, dim3((M + a.waves - 1) / a.waves), dim3(256), lds, as_stream(stream), a);
    return check_launch("stack_vector_pool");
}

extern "C" int pdm_stack_vector_pool_grad(void *stream, const float *grad_new_features, const int *point_cnt_of_grid,
                                          const int *grouped_idxs, float *grad_support_features, int N, int M, int num_c_out,
                                          int num_c_in, int num_total_grids, int num_entries) {
    PDM_REQUIRE(N >= 0 && M >= 0 && num_c_in > 0 && num_total_grids > 0 && num_c_out > 0 && num_c_out % num_total_grids == 0 &&
                    num_entries >= 0,
                PDM_E_BADARG, "stack_vector_pool_grad: bad size");
    const long long total = (long long)num_entries * num_c_in;
    if (total == 0) return 0;
    PDM_REQUIRE(grad_new_features && point_cnt_of_grid && grouped_idxs && grad_support_features, PDM_E_BADARG,
                "stack_vector_pool_grad: null pointer");
    const long long want = (total + 255) / 256;
    hipLaunchKernelGGL(vector_pool_grad_kernel, dim3((unsigned)(want > 65536 ? 65536 : want)), dim3(256), 0, as_stream(stream), total,
                       num_c_in, num_c_out, num_c_out / num_total_grids, num_total_grids, grad_new_features, point_cnt_of_grid,
                       grouped_idxs, grad_support_features);
    return check_launch("stack_vector_pool_grad");
}
