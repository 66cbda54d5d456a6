// three_nn, three_interpolate (+grad) for gfx950.
//
// Semantics: /root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/interpolate_gpu.cu
// (three_nn :16-59, three_interpolate :84-104, grad :127-149).
//
// three_nn: one thread per unknown point; the known set is staged through LDS in coalesced tiles
// as float4 (one ds_read_b128 broadcast per candidate instead of three scalar global loads), the
// three running bests stay in registers.  The reference keeps the bests in double initialised to
// 1e40 and compares a float candidate with strict '<' (:37, :44-55); float bests initialised to
// +inf decide identically (every finite float is < 1e40 and < inf; inf and NaN are < neither) and
// narrow to the same outputs ((float)1e40 == inf).
#include "common.h"

namespace pdm {

constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 2048;

__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(int n, int m,
                                                              const float *__restrict__ unknown,
                                                              const float *__restrict__ known,
                                                              float *__restrict__ dist2,
                                                              int *__restrict__ idx) {
    __shared__ float4 tile[NN_TILE];
    const int b = blockIdx.y;
    const int j = blockIdx.x * NN_THREADS + threadIdx.x;
    const bool have = j < n;
    const float *__restrict__ kn = known + (size_t)b * m * 3;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (have) {
        const float *u = unknown + ((size_t)b * n + j) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    float best1 = INFINITY, best2 = INFINITY, best3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int base = 0; base < m; base += NN_TILE) {
        const int cnt = min(NN_TILE, m - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += NN_THREADS) {
            const float *p = kn + (size_t)(base + i) * 3;
            tile[i] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        if (have) {
            for (int k = 0; k < cnt; ++k) {
                const float4 p = tile[k];
                const float d = sqdist(ux - p.x, uy - p.y, uz - p.z);
                if (d < best1) {
                    best3 = best2; i3 = i2;
                    best2 = best1; i2 = i1;
                    best1 = d; i1 = base + k;
                } else if (d < best2) {
                    best3 = best2; i3 = i2;
                    best2 = d; i2 = base + k;
                } else if (d < best3) {
                    best3 = d; i3 = base + k;
                }
            }
        }
    }
    if (have) {
        float *od = dist2 + ((size_t)b * n + j) * 3;
        int *oi = idx + ((size_t)b * n + j) * 3;
        od[0] = best1; od[1] = best2; od[2] = best3;
        oi[0] = i1; oi[1] = i2; oi[2] = i3;
    }
}

constexpr int TI_THREADS = 256;
constexpr int TI_CG = 8;  // channels per workgroup; idx/weight registers reused across them

// out[b,c,j] = fma(w2,p2, fma(w1,p1, rn(w0*p0)))  (rounding sequence pinned, see oracle)
__global__ __launch_bounds__(TI_THREADS) void three_interpolate_kernel(
    int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out) {
    const int b = blockIdx.z;
    const int j = blockIdx.x * TI_THREADS + threadIdx.x;
    if (j >= n) return;
    const int *id = idx + ((size_t)b * n + j) * 3;
    const float *w = weight + ((size_t)b * n + j) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const int c0 = blockIdx.y * TI_CG;
    const int c1 = min(c0 + TI_CG, c);
    for (int ci = c0; ci < c1; ++ci) {
        const float *__restrict__ row = points + ((size_t)b * c + ci) * m;
        float t = __fmul_rn(w0, row[i0]);
        t = __fmaf_rn(w1, row[i1], t);
        out[((size_t)b * c + ci) * n + j] = __fmaf_rn(w2, row[i2], t);
    }
}

// The same with the known rows of `tc` channels staged in LDS (tc * m floats): the three scattered reads per output are
// LDS reads instead of L1 line fetches (the global form runs at the L1 line rate, ~1 TB/s of output).  Workgroup =
// (channel group, cloud); a thread keeps a point's indices and weights across the group's channels.
constexpr int TIF_THREADS = 512;
__global__ __launch_bounds__(TIF_THREADS) void three_interpolate_lds_kernel(int c, int m, int n, int tc,
                                                                            const float *__restrict__ points,
                                                                            const int *__restrict__ idx,
                                                                            const float *__restrict__ weight,
                                                                            float *__restrict__ out) {
    extern __shared__ float ti_rows[];   // [tc][m]
    const int b = blockIdx.y, c0 = blockIdx.x * tc, nc = min(tc, c - c0);
    const float *__restrict__ src = points + ((size_t)b * c + c0) * m;
    for (int i = threadIdx.x; i < nc * m; i += TIF_THREADS) ti_rows[i] = src[i];   // the group's rows are contiguous
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += TIF_THREADS) {
        const int *id = idx + ((size_t)b * n + j) * 3;
        const float *w = weight + ((size_t)b * n + j) * 3;
        const int i0 = id[0], i1 = id[1], i2 = id[2];
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        for (int ci = 0; ci < nc; ++ci) {
            const float *row = ti_rows + (size_t)ci * m;
            float t = __fmul_rn(w0, row[i0]);
            t = __fmaf_rn(w1, row[i1], t);
            out[((size_t)b * c + c0 + ci) * n + j] = __fmaf_rn(w2, row[i2], t);
        }
    }
}

__global__ __launch_bounds__(TI_THREADS) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const int j = blockIdx.x * TI_THREADS + threadIdx.x;
    if (j >= n) return;
    const int *id = idx + ((size_t)b * n + j) * 3;
    const float *w = weight + ((size_t)b * n + j) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const int c0 = blockIdx.y * TI_CG;
    const int c1 = min(c0 + TI_CG, c);
    for (int ci = c0; ci < c1; ++ci) {
        const float g = grad_out[((size_t)b * c + ci) * n + j];
        float *__restrict__ row = grad_points + ((size_t)b * c + ci) * m;
        atomicAdd(row + i0, g * w0);
        atomicAdd(row + i1, g * w1);
        atomicAdd(row + i2, g * w2);
    }
}

// dist2 (rows,3) -> weight (rows,3): the reference's python glue after three_nn, in one pass and with the same
// correctly-rounded operations in the same order (pointnet2_utils.py:98 sqrt; pointnet2_modules.py:154-156).
__global__ __launch_bounds__(256) void three_nn_weights_kernel(long long rows, const float *__restrict__ dist2,
                                                              float *__restrict__ dist, float *__restrict__ weight) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float d0 = __fsqrt_rn(dist2[r * 3 + 0]), d1 = __fsqrt_rn(dist2[r * 3 + 1]), d2 = __fsqrt_rn(dist2[r * 3 + 2]);
    const float r0 = __fdiv_rn(1.0f, __fadd_rn(d0, 1e-8f)), r1 = __fdiv_rn(1.0f, __fadd_rn(d1, 1e-8f)),
                r2 = __fdiv_rn(1.0f, __fadd_rn(d2, 1e-8f));
    const float norm = __fadd_rn(__fadd_rn(r0, r1), r2);
    weight[r * 3 + 0] = __fdiv_rn(r0, norm); weight[r * 3 + 1] = __fdiv_rn(r1, norm); weight[r * 3 + 2] = __fdiv_rn(r2, norm);
    if (dist) { dist[r * 3 + 0] = d0; dist[r * 3 + 1] = d1; dist[r * 3 + 2] = d2; }
}

// The same backward with the scatter kept on chip: one workgroup owns `tc` channel rows (tc * m floats of LDS) of one
// sample, streams grad_out[b, c, :] coalesced, adds into LDS with ds_add_f32 and writes every grad_points row once.
// The global-atomic form above issues 64 scattered 4-byte atomics per wave instruction (17 ms for FP1 at bs=32);
// this one is bound by the LDS atomic rate.
constexpr int TIL_THREADS = 1024;   // 16 waves per workgroup: the loop is a chain of dependent loads, occupancy hides it
__global__ __launch_bounds__(TIL_THREADS) void three_interpolate_grad_lds_kernel(
    int c, int n, int m, int tc, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
    extern __shared__ float s_acc[];   // tc x m
    const int b = blockIdx.y, c0 = blockIdx.x * tc;
    const int nc = min(tc, c - c0);
    for (int e = threadIdx.x; e < nc * m; e += TIL_THREADS) s_acc[e] = 0.0f;
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += TIL_THREADS) {
        const int *id = idx + ((size_t)b * n + j) * 3;
        const float *w = weight + ((size_t)b * n + j) * 3;
        const int i0 = id[0], i1 = id[1], i2 = id[2];
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        for (int ci = 0; ci < nc; ++ci) {
            const float g = grad_out[((size_t)b * c + c0 + ci) * n + j];
            float *row = s_acc + ci * m;
            atomicAdd(row + i0, g * w0);
            atomicAdd(row + i1, g * w1);
            atomicAdd(row + i2, g * w2);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nc * m; e += TIL_THREADS) {
        const int ci = e / m, k = e - ci * m;
        grad_points[((size_t)b * c + c0 + ci) * m + k] += s_acc[e];   // rows are exclusive to this workgroup
    }
}

// ---- backward of three_interpolate without atomics in the inner loop -------------------------------------------------
// LDS float atomics turned out to be the limit of the kernel above: ~270 GB/s of grad_out for every shape, i.e. one
// ds_add_f32 lane every three cycles per CU.  The scatter is inverted once per (idx, weight) pair instead: per cloud a CSR
// table "known point k <- its (unknown point j, weight) contributions" (counting sort, one workgroup per cloud), then
//   grad_points[b, c, k] += sum_{(j, w) in list(k)} w * grad_out[b, c, j]
// with the grad_out rows of a few channels staged in LDS (coalesced global reads, random LDS READS) and one thread per
// known point.  Needs m <= 16384 (LDS histogram) and n <= 65535 (16-bit j).
constexpr int TIC_THREADS = 1024;

// ne = contributions per cloud, `per` of them per source row (three_interpolate: 3 per unknown point, with weights;
// group_points: 1 per grouped slot, weight 1)
__global__ __launch_bounds__(TIC_THREADS) void interp_csr_build_kernel(int ne, int per, int m, const int *__restrict__ idx,
                                                                       const float *__restrict__ weight,
                                                                       int *__restrict__ start_all,
                                                                       unsigned short *__restrict__ ej_all,
                                                                       float *__restrict__ ew_all) {
    extern __shared__ int s_cnt[];   // m counters, then running cursors
    __shared__ int s_wave[TIC_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int *__restrict__ id = idx + (size_t)b * ne;
    const float *__restrict__ w = weight ? weight + (size_t)b * ne : nullptr;
    int *__restrict__ start = start_all + (size_t)b * (m + 1);
    unsigned short *__restrict__ ej = ej_all + (size_t)b * ne;
    float *__restrict__ ew = ew_all + (size_t)b * ne;
    for (int k = tid; k < m; k += TIC_THREADS) s_cnt[k] = 0;
    __syncthreads();
    for (int e = tid; e < ne; e += TIC_THREADS) {
        const int k = id[e];
        if (k >= 0 && k < m) atomicAdd(&s_cnt[k], 1);
    }
    __syncthreads();
    // exclusive scan of the m counters: contiguous chunk per thread, wave scan, wave offsets
    const int chunk = (m + TIC_THREADS - 1) / TIC_THREADS;
    const int k0 = tid * chunk, k1 = min(k0 + chunk, m);
    int local = 0;
    for (int k = k0; k < k1; ++k) local += s_cnt[k];
    int incl = local;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < wave; ++q) base += s_wave[q];
    int run = base + incl - local;
    for (int k = k0; k < k1; ++k) {
        const int cnt = s_cnt[k];
        start[k] = run;
        s_cnt[k] = run;      // becomes the fill cursor
        run += cnt;
    }
    if (tid == TIC_THREADS - 1) start[m] = run;
    __syncthreads();
    for (int e = tid; e < ne; e += TIC_THREADS) {
        const int k = id[e];
        if (k < 0 || k >= m) continue;
        const int pos = atomicAdd(&s_cnt[k], 1);
        ej[pos] = (unsigned short)(e / per);
        ew[pos] = w ? w[e] : 1.0f;
    }
}

template <int TC>
__global__ __launch_bounds__(TIC_THREADS) void interp_grad_csr_kernel(int c, int n, int ne, int m,
                                                                      const float *__restrict__ grad_out,
                                                                      const int *__restrict__ start_all,
                                                                      const unsigned short *__restrict__ ej_all,
                                                                      const float *__restrict__ ew_all,
                                                                      float *__restrict__ grad_points) {
    extern __shared__ float s_g[];   // TC x n
    const int b = blockIdx.y, c0 = blockIdx.x * TC, tid = threadIdx.x;
    const int nc = min(TC, c - c0);
    const float *__restrict__ g = grad_out + ((size_t)b * c + c0) * n;
    for (int e = tid; e < nc * n; e += TIC_THREADS) s_g[e] = g[e];   // nc consecutive rows are one contiguous block
    __syncthreads();
    const int *__restrict__ start = start_all + (size_t)b * (m + 1);
    const unsigned short *__restrict__ ej = ej_all + (size_t)b * ne;
    const float *__restrict__ ew = ew_all + (size_t)b * ne;
    for (int k = tid; k < m; k += TIC_THREADS) {
        const int s = start[k], e = start[k + 1];
        float acc[TC];
#pragma unroll
        for (int ci = 0; ci < TC; ++ci) acc[ci] = 0.0f;
        for (int p = s; p < e; ++p) {
            const int j = ej[p];
            const float w = ew[p];
#pragma unroll
            for (int ci = 0; ci < TC; ++ci)
                if (ci < nc) acc[ci] += s_g[ci * n + j] * w;
        }
#pragma unroll
        for (int ci = 0; ci < TC; ++ci)
            if (ci < nc) grad_points[((size_t)b * c + c0 + ci) * m + k] += acc[ci];   // rows are exclusive to this workgroup
    }
}

}  // namespace pdm

using namespace pdm;

// Shared by three_interpolate and group_points: rows of `row_len` floats per (cloud, channel) in grad_out, ne = per * rows'
// contributions per cloud scattered onto m targets.  Workspace: csr_workspace_bytes(b, ne, m).
namespace pdm {
size_t csr_workspace_bytes(int b, long long ne, int m) {
    if (b <= 0 || ne <= 0 || m <= 0) return 0;
    return ((size_t)b * (m + 1) * sizeof(int) + 15) / 16 * 16 + ((size_t)b * ne * sizeof(unsigned short) + 15) / 16 * 16 +
           (size_t)b * ne * sizeof(float) + 64;
}
bool csr_form_applies(int b, int row_len, long long ne, int m) {
    return m >= 1 && m <= 16384 && row_len >= 1 && row_len <= 32768 && ne <= 0x7fffffffll / 4 && b <= 65535;
}
int csr_scatter_grad_launch(void *stream, const char *who, int b, int c, int row_len, int per, int m, const float *grad_out,
                            const int *idx, const float *weight, float *grad_points, void *workspace) {
    const int ne = row_len * per;
    uintptr_t p = (reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15;
    int *start = reinterpret_cast<int *>(p);
    p += ((size_t)b * (m + 1) * sizeof(int) + 15) / 16 * 16;
    unsigned short *ej = reinterpret_cast<unsigned short *>(p);
    p += ((size_t)b * ne * sizeof(unsigned short) + 15) / 16 * 16;
    float *ew = reinterpret_cast<float *>(p);
    if ((size_t)m * sizeof(int) + 1024 > 64 * 1024) {   // dynamic + static LDS above the default 64 KB (m near 16384)
        // (the kernel also holds a small static block: dynamic + static must stay within the 160 KB of a CU)
        const int e = grant_lds(reinterpret_cast<const void *>(&interp_csr_build_kernel), 128 * 1024);
        PDM_REQUIRE(e == 0, PDM_E_TOOLARGE, "%s: cannot obtain %zu bytes of LDS", who, (size_t)m * sizeof(int));
    }
    hipLaunchKernelGGL(interp_csr_build_kernel, dim3(b), dim3(TIC_THREADS), (size_t)m * sizeof(int), as_stream(stream), ne, per, m,
                       idx, weight, start, ej, ew);
    int rc = check_launch(who);
    if (rc) return rc;
    // channel rows staged per workgroup: as many as fit 128 KB of LDS, at most 8
    const int tc = row_len <= 4096 ? 8 : row_len <= 8192 ? 4 : row_len <= 16384 ? 2 : 1;
#define PDM_TIC_LAUNCH(TCV, SLOT)                                                                                      \
    do {                                                                                                               \
        const size_t lds = (size_t)TCV * row_len * sizeof(float);                                                      \
        if (lds > 64 * 1024) {   /* granted per function and per device (common.h) */                                  \
            const int e = grant_lds(reinterpret_cast<const void *>(&interp_grad_csr_kernel<TCV>), 160 * 1024);         \
            PDM_REQUIRE(e == 0, PDM_E_TOOLARGE, "%s: cannot obtain %zu bytes of LDS", who, lds);                       \
        }                                                                                                              \
        hipLaunchKernelGGL(interp_grad_csr_kernel<TCV>, dim3(divup(c, TCV), b), dim3(TIC_THREADS), lds, as_stream(stream), \
                           c, row_len, ne, m, grad_out, start, ej, ew, grad_points);                                   \
    } while (0)
    if (tc == 8) PDM_TIC_LAUNCH(8, 0);
    else if (tc == 4) PDM_TIC_LAUNCH(4, 1);
    else if (tc == 2) PDM_TIC_LAUNCH(2, 2);
    else PDM_TIC_LAUNCH(1, 3);
#undef PDM_TIC_LAUNCH
    return check_launch(who);
}
}  // namespace pdm

// ---- rows form of the FP module's input (training path, csrc/train_gemm.hip consumes it) ------------------------------------
// The reference forms cat([three_interpolate(known_feats, idx, weight), unknow_feats], dim=1) as a channel-major fp32 tensor
// (pointnet2_modules.py:158-165) and hands it to the shared MLP.  Here the same values are written ONCE, as the bf16 rows
// (B, n, ld) the MFMA layers read: [ interpolated (C2) | skip (C1) | zeros up to ld ], each element the fp32 value of the
// reference expression (the pinned fma order of three_interpolate_kernel) rounded to nearest even — exactly what autocast's
// cast of the concatenated fp32 tensor holds.  known (B, m, C2) and skip (B, n, C1) are point-major rows, fp32 or bf16.
namespace pdm {

__device__ __forceinline__ float icr_load(const void *p, size_t i, bool bf16) {
    return bf16 ? __uint_as_float((unsigned)static_cast<const unsigned short *>(p)[i] << 16) : static_cast<const float *>(p)[i];
}
__device__ __forceinline__ unsigned short icr_bf16(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(256) void interp_concat_rows_kernel(long long total, int n, int m, int c2, int c1, int ld,
                                                                 const void *__restrict__ known, int known_bf16,
                                                                 const void *__restrict__ skip, int skip_bf16,
                                                                 const int *__restrict__ idx, const float *__restrict__ weight,
                                                                 unsigned short *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / ld;                 // b * n + i
        const int ch = (int)(e - row * ld);
        float v = 0.0f;
        if (ch < c2) {
            const long long b = row / n;
            const int *id = idx + row * 3;
            const float *w = weight + row * 3;
            const float p0 = icr_load(known, ((size_t)b * m + id[0]) * c2 + ch, known_bf16);
            const float p1 = icr_load(known, ((size_t)b * m + id[1]) * c2 + ch, known_bf16);
            const float p2 = icr_load(known, ((size_t)b * m + id[2]) * c2 + ch, known_bf16);
            v = __fmaf_rn(w[2], p2, __fmaf_rn(w[1], p1, __fmul_rn(w[0], p0)));
        } else if (ch < c2 + c1) {
            v = icr_load(skip, (size_t)row * c1 + (ch - c2), skip_bf16);
        }
        out[e] = icr_bf16(v);
    }
}

// The same, eight channels per thread (C2 % 8 == 0, ld % 8 == 0, 16-byte aligned rows): one 16-byte store per thread, the
// three known rows read as 16-byte (bf16) or 2 x 16-byte (fp32) pieces — an eighth of the instructions of the element form.
template <bool KB>
__global__ __launch_bounds__(256) void interp_concat_rows8_kernel(long long total8, int n, int m, int c2, int c1, int ld,
                                                                  const void *__restrict__ known, const void *__restrict__ skip, int skip_bf16,
                                                                  const int *__restrict__ idx, const float *__restrict__ weight,
                                                                  unsigned short *__restrict__ out) {
    const int cpr = ld >> 3;    // chunks per row
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total8; e += (long long)gridDim.x * 256) {
        const long long row = e / cpr;
        const int ch = (int)(e - row * cpr) << 3;
        float v[8];
        if (ch < c2) {
            const long long b = row / n;
            const int *id = idx + row * 3;
            const float *w = weight + row * 3;
            float p[3][8];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const size_t off = ((size_t)b * m + id[k]) * c2 + ch;
                if (KB) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(static_cast<const unsigned short *>(known) + off);
                    const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) { p[k][2 * t] = __uint_as_float(u[t] << 16); p[k][2 * t + 1] = __uint_as_float(u[t] & 0xffff0000u); }
                } else {
                    const float4 a = *reinterpret_cast<const float4 *>(static_cast<const float *>(known) + off);
                    const float4 c = *reinterpret_cast<const float4 *>(static_cast<const float *>(known) + off + 4);
                    p[k][0] = a.x; p[k][1] = a.y; p[k][2] = a.z; p[k][3] = a.w; p[k][4] = c.x; p[k][5] = c.y; p[k][6] = c.z; p[k][7] = c.w;
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = __fmaf_rn(w[2], p[2][t], __fmaf_rn(w[1], p[1][t], __fmul_rn(w[0], p[0][t])));
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int c = ch + t - c2;
                v[t] = c < c1 ? icr_load(skip, (size_t)row * c1 + c, skip_bf16) : 0.0f;
            }
        }
        uint4 o;
        o.x = (unsigned)icr_bf16(v[0]) | ((unsigned)icr_bf16(v[1]) << 16); o.y = (unsigned)icr_bf16(v[2]) | ((unsigned)icr_bf16(v[3]) << 16);
        o.z = (unsigned)icr_bf16(v[4]) | ((unsigned)icr_bf16(v[5]) << 16); o.w = (unsigned)icr_bf16(v[6]) | ((unsigned)icr_bf16(v[7]) << 16);
        *reinterpret_cast<uint4 *>(out + e * 8) = o;
    }
}

// d known[b, j, c] = sum over the CSR list of j of w * dx[b, i, c], c < C2: one wave per known point, lanes over channels,
// every gradient row read contiguously, no atomics (fixed order: reproducible)
template <bool OB>   // OB: dknown is written as bf16 (round to nearest even of the same fp32 sums) instead of fp32
__global__ __launch_bounds__(256) void interp_rows_grad_kernel(int n, int m, int c2, int ld, int ne, const unsigned short *__restrict__ dx,
                                                               const int *__restrict__ start_all, const unsigned short *__restrict__ ej_all,
                                                               const float *__restrict__ ew_all, void *__restrict__ dknown_v) {
    float *__restrict__ dknown = static_cast<float *>(dknown_v);
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int *__restrict__ start = start_all + (size_t)b * (m + 1);
    const unsigned short *__restrict__ ej = ej_all + (size_t)b * ne;
    const float *__restrict__ ew = ew_all + (size_t)b * ne;
    if ((c2 & 7) == 0 && (ld & 7) == 0) {        // eight channels per lane: 16-byte reads of the gradient rows
        // lpp lanes per known point (C2 / 8 rounded up to a power of two, <= 64): at C2 = 256 a wave serves two points instead of
        // leaving half its lanes idle; four list entries are in flight per step (one dependent index + weight + row load at a
        // time was a latency chain).  Same summation order per channel.
        int lpp = 1;
        while (lpp < (c2 >> 3) && lpp < 64) lpp <<= 1;
        const int ppw = 64 / lpp;
        const int j = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ppw + lane / lpp;
        if (j >= m) return;
        const int s = start[j], e = start[j + 1];
        const unsigned short *__restrict__ dxb = dx + (size_t)b * n * ld;
        auto fma8 = [](float (&acc)[8], float w, uint4 q) {
            const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[2 * t] = __fmaf_rn(w, __uint_as_float(u[t] << 16), acc[2 * t]);
                acc[2 * t + 1] = __fmaf_rn(w, __uint_as_float(u[t] & 0xffff0000u), acc[2 * t + 1]);
            }
        };
        for (int ch = (lane % lpp) * 8; ch < c2; ch += lpp * 8) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int p = s;
            for (; p + 4 <= e; p += 4) {
                const float w0 = ew[p], w1 = ew[p + 1], w2 = ew[p + 2], w3 = ew[p + 3];
                const uint4 q0 = *reinterpret_cast<const uint4 *>(dxb + (size_t)ej[p] * ld + ch),
                            q1 = *reinterpret_cast<const uint4 *>(dxb + (size_t)ej[p + 1] * ld + ch),
                            q2 = *reinterpret_cast<const uint4 *>(dxb + (size_t)ej[p + 2] * ld + ch),
                            q3 = *reinterpret_cast<const uint4 *>(dxb + (size_t)ej[p + 3] * ld + ch);
                fma8(acc, w0, q0); fma8(acc, w1, q1); fma8(acc, w2, q2); fma8(acc, w3, q3);
            }
            for (; p < e; ++p) fma8(acc, ew[p], *reinterpret_cast<const uint4 *>(dxb + (size_t)ej[p] * ld + ch));
            if constexpr (OB) {
                uint4 o;
                o.x = (unsigned)icr_bf16(acc[0]) | ((unsigned)icr_bf16(acc[1]) << 16); o.y = (unsigned)icr_bf16(acc[2]) | ((unsigned)icr_bf16(acc[3]) << 16);
                o.z = (unsigned)icr_bf16(acc[4]) | ((unsigned)icr_bf16(acc[5]) << 16); o.w = (unsigned)icr_bf16(acc[6]) | ((unsigned)icr_bf16(acc[7]) << 16);
                *reinterpret_cast<uint4 *>(static_cast<unsigned short *>(dknown_v) + ((size_t)b * m + j) * c2 + ch) = o;
            } else {
                float *o = dknown + ((size_t)b * m + j) * c2 + ch;
                *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4 *>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
        }
        return;
    }
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= m) return;
    const int s = start[j], e = start[j + 1];
    for (int c0 = 0; c0 < c2; c0 += 64) {
        const int ch = c0 + lane;
        if (ch < c2) {
            float acc = 0.0f;
            for (int p = s; p < e; ++p)
                acc = __fmaf_rn(ew[p], __uint_as_float((unsigned)dx[((size_t)b * n + ej[p]) * ld + ch] << 16), acc);
            if constexpr (OB) static_cast<unsigned short *>(dknown_v)[((size_t)b * m + j) * c2 + ch] = icr_bf16(acc);
            else dknown[((size_t)b * m + j) * c2 + ch] = acc;
        }
    }
}

}  // namespace pdm

// out (B, n, ld) bf16 = [ three_interpolate(known, idx, weight) | skip | 0 ], ld >= C2 + C1.  known (B, m, C2), skip (B, n, C1)
// (may be null when C1 == 0) are point-major rows, fp32 or bf16 (known_bf16 / skip_bf16); idx, weight (B, n, 3).
extern "C" int pdm_interp_concat_rows(void *stream, int b, int n, int m, int c2, int c1, int ld, const void *known, int known_bf16,
                                      const void *skip, int skip_bf16, const int *idx, const float *weight, void *out) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c2 >= 0 && c1 >= 0 && ld >= c2 + c1, PDM_E_BADARG, "interp_concat_rows: bad size");
    const long long total = (long long)b * n * ld;
    if (total == 0) return 0;
    PDM_REQUIRE(out && (c2 == 0 || (known && idx && weight && m >= 1)) && (c1 == 0 || skip), PDM_E_BADARG, "interp_concat_rows: null pointer");
    if (c2 % 8 == 0 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(known) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const long long total8 = total / 8;
        const long long want8 = (total8 + 255) / 256;
        const int blocks8 = (int)(want8 < 256 * 64 ? want8 : 256 * 64);
        if (known_bf16)
            hipLaunchKernelGGL(interp_concat_rows8_kernel<true>, dim3(blocks8), dim3(256), 0, as_stream(stream), total8, n, m, c2, c1, ld, known,
                               skip, skip_bf16, idx, weight, static_cast<unsigned short *>(out));
        else
            hipLaunchKernelGGL(interp_concat_rows8_kernel<false>, dim3(blocks8), dim3(256), 0, as_stream(stream), total8, n, m, c2, c1, ld, known,
                               skip, skip_bf16, idx, weight, static_cast<unsigned short *>(out));
        return check_launch("interp_concat_rows");
    }
    const long long want = (total + 255) / 256;
    const int blocks = (int)(want < 256 * 64 ? want : 256 * 64);
    hipLaunchKernelGGL(interp_concat_rows_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), total, n, m, c2, c1, ld, known, known_bf16,
                       skip, skip_bf16, idx, weight, static_cast<unsigned short *>(out));
    return check_launch("interp_concat_rows");
}

// Its backward towards the known features: dx (B, n, ld) bf16 -> dknown (B, m, C2) fp32, fully written.  The skip features'
// gradient is the column block dx[..., C2 : C2 + C1] itself.  workspace: pdm_three_interpolate_grad_ws_bytes(b, n, m).
extern "C" int pdm_interp_concat_rows_grad_out(void *stream, int b, int n, int m, int c2, int ld, const void *dx, const int *idx,
                                               const float *weight, void *dknown, int out_bf16, void *workspace, size_t workspace_bytes);
extern "C" int pdm_interp_concat_rows_grad(void *stream, int b, int n, int m, int c2, int ld, const void *dx, const int *idx,
                                           const float *weight, float *dknown, void *workspace, size_t workspace_bytes) {
    return pdm_interp_concat_rows_grad_out(stream, b, n, m, c2, ld, dx, idx, weight, dknown, 0, workspace, workspace_bytes);
}
// the same with the gradient of the known rows written as bf16 (out_bf16 = 1: the rounding a cast of the fp32 result would do)
extern "C" int pdm_interp_concat_rows_grad_out(void *stream, int b, int n, int m, int c2, int ld, const void *dx, const int *idx,
                                               const float *weight, void *dknown, int out_bf16, void *workspace, size_t workspace_bytes) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c2 >= 0 && ld >= c2, PDM_E_BADARG, "interp_concat_rows_grad: bad size");
    if (b == 0 || m == 0 || c2 == 0) return 0;
    PDM_REQUIRE(dknown && workspace && (n == 0 || (dx && idx && weight)), PDM_E_BADARG, "interp_concat_rows_grad: null pointer");
    PDM_REQUIRE(m <= 16384 && n <= 65535 && b <= 65535, PDM_E_TOOLARGE, "interp_concat_rows_grad: m=%d (<= 16384), n=%d (<= 65535)", m, n);
    PDM_REQUIRE(workspace_bytes >= csr_workspace_bytes(b, 3ll * n, m), PDM_E_BADARG, "interp_concat_rows_grad: workspace of %zu bytes, need %zu",
                workspace_bytes, csr_workspace_bytes(b, 3ll * n, m));
    const int ne = 3 * n;
    uintptr_t p = (reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15;
    int *start = reinterpret_cast<int *>(p);
    p += ((size_t)b * (m + 1) * sizeof(int) + 15) / 16 * 16;
    unsigned short *ej = reinterpret_cast<unsigned short *>(p);
    p += ((size_t)b * ne * sizeof(unsigned short) + 15) / 16 * 16;
    float *ew = reinterpret_cast<float *>(p);
    if ((size_t)m * sizeof(int) + 1024 > 64 * 1024) {
        const int e = grant_lds(reinterpret_cast<const void *>(&interp_csr_build_kernel), 128 * 1024);
        PDM_REQUIRE(e == 0, PDM_E_TOOLARGE, "interp_concat_rows_grad: cannot obtain %zu bytes of LDS", (size_t)m * sizeof(int));
    }
    hipLaunchKernelGGL(interp_csr_build_kernel, dim3(b), dim3(TIC_THREADS), (size_t)m * sizeof(int), as_stream(stream), ne, 3, m, idx, weight,
                       start, ej, ew);
    int rc = check_launch("interp_concat_rows_grad(csr)");
    if (rc) return rc;
    int lpp = 1;     // as in the kernel's eight-channel form: points per workgroup = 4 * 64 / lpp
    if ((c2 & 7) == 0 && (ld & 7) == 0)
        while (lpp < (c2 >> 3) && lpp < 64) lpp <<= 1;
    else lpp = 64;
    const int ppb = 4 * (64 / lpp);
    if (out_bf16)
        hipLaunchKernelGGL(interp_rows_grad_kernel<true>, dim3((unsigned)((m + ppb - 1) / ppb), (unsigned)b), dim3(256), 0, as_stream(stream), n, m, c2, ld, ne,
                           static_cast<const unsigned short *>(dx), start, ej, ew, dknown);
    else
        hipLaunchKernelGGL(interp_rows_grad_kernel<false>, dim3((unsigned)((m + ppb - 1) / ppb), (unsigned)b), dim3(256), 0, as_stream(stream), n, m, c2, ld, ne,
                           static_cast<const unsigned short *>(dx), start, ej, ew, dknown);
    return check_launch("interp_concat_rows_grad");
}

extern "C" size_t pdm_three_interpolate_grad_ws_bytes(int b, int n, int m) { return csr_workspace_bytes(b, 3ll * n, m); }

// pdm_three_interpolate_grad with a caller-provided workspace (pdm_three_interpolate_grad_ws_bytes(b, n, m) bytes): the
// scatter is inverted into per-cloud CSR lists first, the accumulation then runs without atomics (same sums, different
// fp32 summation order).  m <= 16384 and n <= 32768 (16-bit row numbers, rows staged in LDS), otherwise the plain entry point.
extern "C" int pdm_three_interpolate_grad_ws(void *stream, int b, int c, int n, int m, const float *grad_out,
                                             const int *idx, const float *weight, float *grad_points, void *workspace,
                                             size_t workspace_bytes) {
    PDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, PDM_E_BADARG, "three_interpolate_grad_ws: negative size");
    if (b == 0 || c == 0 || n == 0) return 0;
    PDM_REQUIRE(grad_out && idx && weight && grad_points, PDM_E_BADARG, "three_interpolate_grad_ws: null pointer");
    if (!csr_form_applies(b, n, 3ll * n, m))
        return pdm_three_interpolate_grad(stream, b, c, n, m, grad_out, idx, weight, grad_points);
    PDM_REQUIRE(workspace && workspace_bytes >= pdm_three_interpolate_grad_ws_bytes(b, n, m), PDM_E_BADARG,
                "three_interpolate_grad_ws: workspace of %zu bytes, need %zu", workspace_bytes,
                pdm_three_interpolate_grad_ws_bytes(b, n, m));
    return csr_scatter_grad_launch(stream, "three_interpolate_grad_ws", b, c, n, 3, m, grad_out, idx, weight, grad_points, workspace);
}

extern "C" int pdm_three_nn_weights(void *stream, long long rows, const float *dist2, float *dist, float *weight) {
    PDM_REQUIRE(rows >= 0, PDM_E_BADARG, "three_nn_weights: negative size");
    if (rows == 0) return 0;
    PDM_REQUIRE(dist2 && weight, PDM_E_BADARG, "three_nn_weights: null pointer");
    PDM_REQUIRE(rows <= 0x7fffffffll * 256, PDM_E_TOOLARGE, "three_nn_weights: too many rows");
    hipLaunchKernelGGL(three_nn_weights_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), rows,
                       dist2, dist, weight);
    return check_launch("three_nn_weights");
}

extern "C" int pdm_three_nn(void *stream, int b, int n, int m, const float *unknown,
                            const float *known, float *dist2, int *idx) {
    PDM_REQUIRE(b >= 0 && n >= 0 && m >= 0, PDM_E_BADARG, "three_nn: negative size");
    if (b == 0 || n == 0) return 0;
    PDM_REQUIRE(unknown && dist2 && idx && (m == 0 || known), PDM_E_BADARG, "three_nn: null pointer");
    PDM_REQUIRE(b <= 65535, PDM_E_TOOLARGE, "three_nn: b=%d exceeds grid", b);
    dim3 grid(divup(n, NN_THREADS), b);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(NN_THREADS), 0, as_stream(stream), n, m, unknown,
                       known, dist2, idx);
    return check_launch("three_nn");
}

extern "C" int pdm_three_interpolate(void *stream, int b, int c, int m, int n, const float *points,
                                     const int *idx, const float *weight, float *out) {
    PDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, PDM_E_BADARG, "three_interpolate: negative size");
    if (b == 0 || c == 0 || n == 0) return 0;
    PDM_REQUIRE(points && idx && weight && out, PDM_E_BADARG, "three_interpolate: null pointer");
    PDM_REQUIRE(b <= 65535 && divup(c, TI_CG) <= 65535, PDM_E_TOOLARGE, "three_interpolate: exceeds grid");
    if (m >= 1 && m <= 8192 && n >= 4 * TIF_THREADS && c >= 2) {
        // rows of up to 64 KB of known points per workgroup (>= 2 workgroups per CU), enough groups to fill the chip
        int tc = 16384 / m;
        tc = tc > 16 ? 16 : tc;
        tc = tc > c ? c : tc;
        while (tc > 1 && (long long)divup(c, tc) * b < 512) tc >>= 1;
        dim3 grid(divup(c, tc), b);
        hipLaunchKernelGGL(three_interpolate_lds_kernel, grid, dim3(TIF_THREADS), (size_t)tc * m * sizeof(float), as_stream(stream),
                           c, m, n, tc, points, idx, weight, out);
        return check_launch("three_interpolate");
    }
    dim3 grid(divup(n, TI_THREADS), divup(c, TI_CG), b);
    hipLaunchKernelGGL(three_interpolate_kernel, grid, dim3(TI_THREADS), 0, as_stream(stream), c, m,
                       n, points, idx, weight, out);
    return check_launch("three_interpolate");
}

extern "C" int pdm_three_interpolate_grad(void *stream, int b, int c, int n, int m,
                                          const float *grad_out, const int *idx,
                                          const float *weight, float *grad_points) {
    PDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, PDM_E_BADARG, "three_interpolate_grad: negative size");
    if (b == 0 || c == 0 || n == 0) return 0;
    PDM_REQUIRE(grad_out && idx && weight && grad_points, PDM_E_BADARG, "three_interpolate_grad: null pointer");
    PDM_REQUIRE(b <= 65535 && divup(c, TI_CG) <= 65535, PDM_E_TOOLARGE, "three_interpolate_grad: exceeds grid");
    if (m >= 1 && m <= 16384) {
        // known-point rows of a few channels fit LDS: accumulate there (LDS atomics), write each row once
        int tc = 16384 / m;
        tc = tc > 8 ? 8 : tc;
        tc = tc > c ? c : tc;
        dim3 grid(divup(c, tc), b);
        hipLaunchKernelGGL(three_interpolate_grad_lds_kernel, grid, dim3(TIL_THREADS), (size_t)tc * m * sizeof(float),
                           as_stream(stream), c, n, m, tc, grad_out, idx, weight, grad_points);
        return check_launch("three_interpolate_grad");
    }
    dim3 grid(divup(n, TI_THREADS), divup(c, TI_CG), b);
    hipLaunchKernelGGL(three_interpolate_grad_kernel, grid, dim3(TI_THREADS), 0, as_stream(stream), c,
                       n, m, grad_out, idx, weight, grad_points);
    return check_launch("three_interpolate_grad");
}
