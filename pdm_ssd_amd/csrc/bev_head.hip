// Depthwise 3x3 convolution over a channels-last BEV map, BatchNorm (folded) and ReLU fused:
//   out[b, y, x, c] = relu(sum_{dy,dx in -1..1} w[(dy+1)*3 + dx+1][c] * in[b, y+dy, x+dx, c] + shift[c]),  zero padding.
// The "context learning" stage of the BEV heat-map head (pdm_ssd_amd/dense_heads/pdm_heatmap_head.py; build-defined,
// the reference snapshot holds no source for PDM-SSD's head: SURVEY.md F1).  The PDM neck's grid is channels-last
// storage (B, H, W, C), so a cell is one contiguous row of C floats: a thread owns four channels of one cell (of a strip
// of four cells when the map row holds eight or more), a workgroup a run of cells; the nine taps of neighbouring cells hit L1 / L2 (three map rows = 270 KB per cloud), so
// HBM sees the map once in and once out.  Bound: HBM (8 * C bytes per cell).
#include "common.h"

namespace pdm {

// A map may be held in bf16 (training under bf16 autocast: the convolution's OUTPUT and the gradient that comes back for it; the
// arithmetic stays fp32, one rounding to nearest even on the way out): four channels = one 8-byte load / store.
typedef __bf16 dw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
template <bool BF>
__device__ __forceinline__ float4 dw_ld(const void *__restrict__ base, size_t q) {
    if constexpr (BF) {
        const uint2 v = reinterpret_cast<const uint2 *>(base)[q];
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
    } else {
        return reinterpret_cast<const float4 *>(base)[q];
    }
}
template <bool BF>
__device__ __forceinline__ void dw_st(void *__restrict__ base, size_t q, const float4 &a) {
    if constexpr (BF) {
        const dw_f32x2 lo = {a.x, a.y}, hi = {a.z, a.w};
        uint2 v;
        v.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, dw_bf16x2));
        v.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, dw_bf16x2));
        reinterpret_cast<uint2 *>(base)[q] = v;
    } else {
        reinterpret_cast<float4 *>(base)[q] = a;
    }
}

template <bool IB, bool OB>
__global__ __launch_bounds__(256) void depthwise3x3_cl_kernel(int H, int W, int C4, const void *__restrict__ in,
                                                             const float4 *__restrict__ w, const float4 *__restrict__ shift,
                                                             void *__restrict__ out, int relu) {
    const int b = blockIdx.y;
    const long long cell0 = (long long)blockIdx.x * (256 / C4);
    const int cq = threadIdx.x % C4, lc = threadIdx.x / C4;
    const long long cell = cell0 + lc;
    if (cell >= (long long)H * W || lc >= 256 / C4) return;
    const int y = (int)(cell / W), x = (int)(cell - (long long)y * W);
    const size_t img0 = (size_t)b * H * W * C4;
    float4 acc = shift[cq];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const float4 v = dw_ld<IB>(in, img0 + ((size_t)yy * W + xx) * C4 + cq);
            const float4 k = w[((dy + 1) * 3 + dx + 1) * C4 + cq];
            acc.x = fmaf(k.x, v.x, acc.x); acc.y = fmaf(k.y, v.y, acc.y);
            acc.z = fmaf(k.z, v.z, acc.z); acc.w = fmaf(k.w, v.w, acc.w);
        }
    }
    if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    dw_st<OB>(out, ((size_t)b * H * W + cell) * C4 + cq, acc);
}

// The same, a thread owning a STRIP of four consecutive cells of a map row (same channel quad): the 3 x 6 input cells it needs
// are loaded once (18 loads for 4 outputs instead of 36; the one-cell form ran at 2.5 TB/s of map traffic, bound by the vector
// L1's request rate).  Same fma order per output as the one-cell kernel (taps row by row, left to right): identical results.
template <bool IB, bool OB>
__global__ __launch_bounds__(256) void depthwise3x3_cl_strip_kernel(int H, int W, int C4, const void *__restrict__ in,
                                                                   const float4 *__restrict__ w, const float4 *__restrict__ shift,
                                                                   void *__restrict__ out, int relu, unsigned wg_per_img) {
    // one-dimensional grid of (image, group of strips).  Launch ids go round-robin over the 8 XCDs; XCD x takes a CONTIGUOUS
    // range of work ids, i.e. bands of map rows: the rows above and below a strip then sit in the same L2 (in launch order the
    // three readers of a map row were on three XCDs, and the map crossed the fabric ~3 times: 3.7 TB/s of map traffic).
    const unsigned total = gridDim.x, q_ = total >> 3, r_ = total & 7u, xcd = blockIdx.x & 7u, i_ = blockIdx.x >> 3;
    const unsigned wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + i_;
    const int b = (int)(wg / wg_per_img);
    const unsigned sx = wg % wg_per_img;
    const int spr = (W + 3) / 4;                       // strips per map row
    const int cq = threadIdx.x % C4, ls = threadIdx.x / C4;
    const long long strip = (long long)sx * (256 / C4) + ls;
    if (strip >= (long long)H * spr || ls >= 256 / C4) return;
    const int y = (int)(strip / spr), x0 = (int)(strip - (long long)y * spr) * 4;
    const size_t img0 = (size_t)b * H * W * C4;
    float4 k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = w[t * C4 + cq];
    const float4 sh = shift[cq];
    float4 acc[4] = {sh, sh, sh, sh};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        float4 v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int xx = x0 - 1 + j;
            v[j] = (xx >= 0 && xx < W) ? dw_ld<IB>(in, img0 + ((size_t)yy * W + xx) * C4 + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if (x0 + i + dx - 1 < 0 || x0 + i + dx - 1 >= W) continue;     // the one-cell kernel skips out-of-map taps: same sum
                const float4 kk = k[(dy + 1) * 3 + dx], vv = v[i + dx];
                acc[i].x = fmaf(kk.x, vv.x, acc[i].x); acc[i].y = fmaf(kk.y, vv.y, acc[i].y);
                acc[i].z = fmaf(kk.z, vv.z, acc[i].z); acc[i].w = fmaf(kk.w, vv.w, acc[i].w);
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (x0 + i >= W) break;
        float4 a = acc[i];
        if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
        dw_st<OB>(out, ((size_t)b * H * W + (size_t)y * W + x0 + i) * C4 + cq, a);
    }
}

}  // namespace pdm

// in / out (B, H, W, C) fp32 channels-last (distinct buffers), w (9, C) tap-major with the BatchNorm scale folded in,
// shift (C).  C a multiple of 4, C <= 1024; all pointers 16-byte aligned.
static int bev_depthwise3x3_impl(const char *who, void *stream, int B, int H, int W, int C, const void *in, int in_bf16, const float *w,
                                 const float *shift, void *out, int out_bf16, int relu) {
    using namespace pdm;
    PDM_REQUIRE(B >= 0 && H >= 0 && W >= 0 && C >= 0, PDM_E_BADARG, "%s: negative size", who);
    if (B == 0 || H == 0 || W == 0 || C == 0) return 0;
    PDM_REQUIRE(C % 4 == 0 && C <= 1024 && B <= 65535, PDM_E_BADARG, "%s: C=%d (multiple of 4, <= 1024), B=%d", who, C, B);
    PDM_REQUIRE(in && w && shift && out && in != out, PDM_E_BADARG, "%s: null or aliased pointer", who);
    PDM_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(shift) |
                  reinterpret_cast<uintptr_t>(out)) & 15) == 0, PDM_E_BADARG, "%s: buffers must be 16-byte aligned", who);
    PDM_REQUIRE(!(in_bf16 && out_bf16), PDM_E_BADARG, "%s: bf16 on one side only (forward: fp32 -> bf16, data gradient: bf16 -> fp32)", who);
    const int C4 = C / 4, cells_per_wg = 256 / C4 > 0 ? 256 / C4 : 1;
    PDM_REQUIRE(C4 <= 256, PDM_E_BADARG, "%s: C=%d", who, C);
    const long long cells = (long long)H * W;
    const float4 *w4 = reinterpret_cast<const float4 *>(w), *s4 = reinterpret_cast<const float4 *>(shift);
    if (W >= 8) {   // strips of four cells along x
        const long long strips = (long long)H * ((W + 3) / 4);
        const unsigned wpi = (unsigned)((strips + cells_per_wg - 1) / cells_per_wg);
        const dim3 g(wpi * (unsigned)B), t(256);
        if (in_bf16) hipLaunchKernelGGL((depthwise3x3_cl_strip_kernel<true, false>), g, t, 0, as_stream(stream), H, W, C4, in, w4, s4, out, relu, wpi);
        else if (out_bf16) hipLaunchKernelGGL((depthwise3x3_cl_strip_kernel<false, true>), g, t, 0, as_stream(stream), H, W, C4, in, w4, s4, out, relu, wpi);
        else hipLaunchKernelGGL((depthwise3x3_cl_strip_kernel<false, false>), g, t, 0, as_stream(stream), H, W, C4, in, w4, s4, out, relu, wpi);
        return check_launch(who);
    }
    const dim3 g((unsigned)((cells + cells_per_wg - 1) / cells_per_wg), B), t(256);
    if (in_bf16) hipLaunchKernelGGL((depthwise3x3_cl_kernel<true, false>), g, t, 0, as_stream(stream), H, W, C4, in, w4, s4, out, relu);
    else if (out_bf16) hipLaunchKernelGGL((depthwise3x3_cl_kernel<false, true>), g, t, 0, as_stream(stream), H, W, C4, in, w4, s4, out, relu);
    else hipLaunchKernelGGL((depthwise3x3_cl_kernel<false, false>), g, t, 0, as_stream(stream), H, W, C4, in, w4, s4, out, relu);
    return check_launch(who);
}

extern "C" int pdm_bev_depthwise3x3(void *stream, int B, int H, int W, int C, const float *in, const float *w,
                                    const float *shift, float *out, int relu) {
    return bev_depthwise3x3_impl("bev_depthwise3x3", stream, B, H, W, C, in, 0, w, shift, out, 0, relu);
}

// The same with ONE side of the map in bf16 (training under bf16 autocast): in_bf16 = the input is bf16 (the data gradient, fed the
// bf16 gradient of the output), out_bf16 = the output is rounded to bf16 (the forward).  fp32 arithmetic either way.
extern "C" int pdm_bev_depthwise3x3_t(void *stream, int B, int H, int W, int C, const void *in, int in_bf16, const float *w,
                                      const float *shift, void *out, int out_bf16, int relu) {
    return bev_depthwise3x3_impl("bev_depthwise3x3_t", stream, B, H, W, C, in, in_bf16, w, shift, out, out_bf16, relu);
}

// Weight gradient of the depthwise 3x3 convolution above:
//   gw[tap][c] = sum over (b, y, x) of gout[b, y, x, c] * in[b, y + dy, x + dx, c]      (zero padding)
// A thread owns four channels and walks strips of four cells with 36 running sums; a workgroup's threads with the same
// channel quad are reduced through LDS and added to gw (9, C) with one float atomic per (tap, channel) per workgroup.
// (The data gradient is the same convolution with the taps mirrored: pdm_bev_depthwise3x3 on gout.)
namespace pdm {

template <bool GB>
__global__ __launch_bounds__(256) void depthwise3x3_cl_wgrad_kernel(int H, int W, int C4, long long cells_total, int cells_per_wg,
                                                                   const float4 *__restrict__ in, const void *__restrict__ gout,
                                                                   float *__restrict__ gw) {
    extern __shared__ float red[];   // (256 / C4) rows x 36 x C4 x 4 floats -> reduced over rows
    const int cq = threadIdx.x % C4, lc = threadIdx.x / C4, rows = 256 / C4;
    // (work ranges dealt so that XCD x takes a contiguous stretch of the map: neighbouring ranges share halo rows)
    const unsigned total_ = gridDim.x, q_ = total_ >> 3, r_ = total_ & 7u, xcd_ = blockIdx.x & 7u, i_ = blockIdx.x >> 3;
    const unsigned wg_ = (xcd_ < r_ ? xcd_ * (q_ + 1) : r_ * (q_ + 1) + (xcd_ - r_) * q_) + i_;
    const long long c_begin = (long long)wg_ * cells_per_wg;
    const long long c_end = c_begin + cells_per_wg < cells_total ? c_begin + cells_per_wg : cells_total;
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lc < rows) {
        // units are strips of four cells of one map row: the 3 x 6 input cells around a strip are loaded once (22 loads per
        // four cells instead of 40)
        const int spr = (W + 3) / 4;
        for (long long strip = c_begin + lc; strip < c_end; strip += rows) {
            const long long by = strip / spr;                     // b * H + y
            const int x0 = (int)(strip - by * spr) * 4;
            const int y = (int)(by % H);
            float4 g[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                g[i] = x0 + i < W ? dw_ld<GB>(gout, (size_t)((by * W + x0 + i) * C4 + cq)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                float4 v[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int xx = x0 - 1 + j;
                    v[j] = (xx >= 0 && xx < W) ? in[((by + dy) * W + xx) * C4 + cq] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float4 &a = acc[(dy + 1) * 3 + dx];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 vv = v[i + dx];
                        a.x = fmaf(g[i].x, vv.x, a.x); a.y = fmaf(g[i].y, vv.y, a.y);
                        a.z = fmaf(g[i].z, vv.z, a.z); a.w = fmaf(g[i].w, vv.w, a.w);
                    }
                }
            }
        }
    }
    // reduce over the rows of the workgroup: red[lc][t][cq] (float4)
    float4 *r4 = reinterpret_cast<float4 *>(red);
    if (lc < rows)
#pragma unroll
        for (int t = 0; t < 9; ++t) r4[(lc * 9 + t) * C4 + cq] = acc[t];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * C4 * 4; i += 256) {   // i = (t * C4 + cq) * 4 + comp
        float s = 0.f;
        const int tc = i >> 2, comp = i & 3;
        for (int r = 0; r < rows; ++r) s += red[((r * 9 * C4) + tc) * 4 + comp];
        atomicAdd(gw + i, s);
    }
}

}  // namespace pdm

// in, gout (B, H, W, C) fp32 channels-last; gw (9, C) fp32, ZEROED by the caller, receives the sums (float atomics
// between workgroups: the order of the last additions is free, compare with a tolerance).
static int bev_depthwise3x3_wgrad_impl(const char *who, void *stream, int B, int H, int W, int C, const float *in, const void *gout,
                                       int gout_bf16, float *gw) {
    using namespace pdm;
    PDM_REQUIRE(B >= 0 && H >= 0 && W >= 0 && C >= 0, PDM_E_BADARG, "%s: negative size", who);
    if (B == 0 || H == 0 || W == 0 || C == 0) return 0;
    PDM_REQUIRE(C % 4 == 0 && C <= 1024, PDM_E_BADARG, "%s: C=%d (multiple of 4, <= 1024)", who, C);
    PDM_REQUIRE(in && gout && gw, PDM_E_BADARG, "%s: null pointer", who);
    PDM_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(gout) | reinterpret_cast<uintptr_t>(gw)) & 15) == 0,
                PDM_E_BADARG, "%s: buffers must be 16-byte aligned", who);
    const int C4 = C / 4;
    const long long cells = (long long)B * H * ((W + 3) / 4);   // work units: strips of four cells along x
    const int rows = 256 / C4 > 0 ? 256 / C4 : 1;
    PDM_REQUIRE(C4 <= 256, PDM_E_BADARG, "%s: C=%d", who, C);
    int wgs = 2048;
    if (cells < wgs * (long long)rows) wgs = (int)((cells + rows - 1) / rows);
    const int per = (int)((cells + wgs - 1) / wgs);
    const size_t lds = (size_t)rows * 9 * C4 * 4 * sizeof(float);
    PDM_REQUIRE(lds <= 64 * 1024, PDM_E_TOOLARGE, "%s: %zu bytes of LDS", who, lds);
    const dim3 g((unsigned)((cells + per - 1) / per)), t(256);
    if (gout_bf16) hipLaunchKernelGGL((depthwise3x3_cl_wgrad_kernel<true>), g, t, lds, as_stream(stream), H, W, C4, cells, per,
                                      reinterpret_cast<const float4 *>(in), gout, gw);
    else hipLaunchKernelGGL((depthwise3x3_cl_wgrad_kernel<false>), g, t, lds, as_stream(stream), H, W, C4, cells, per,
                            reinterpret_cast<const float4 *>(in), gout, gw);
    return check_launch(who);
}

extern "C" int pdm_bev_depthwise3x3_wgrad(void *stream, int B, int H, int W, int C, const float *in, const float *gout, float *gw) {
    return bev_depthwise3x3_wgrad_impl("bev_depthwise3x3_wgrad", stream, B, H, W, C, in, gout, 0, gw);
}

// the same with the output gradient held in bf16 (gout_bf16 = 1): see pdm_bev_depthwise3x3_t
extern "C" int pdm_bev_depthwise3x3_wgrad_t(void *stream, int B, int H, int W, int C, const float *in, const void *gout, int gout_bf16,
                                            float *gw) {
    return bev_depthwise3x3_wgrad_impl("bev_depthwise3x3_wgrad_t", stream, B, H, W, C, in, gout, gout_bf16, gw);
}

// Point head epilogue in one pass: class scores and decoded boxes of every point.
//   score[i]  = sigmoid(max_c cls[i][c])                                  (point_head_box.py:97-98)
//   box[i]    = PointResidualCoder.decode(code[i], point[i], argmax_c + 1)  (box_coder_utils.py:188-222, use_mean_size):
//               x, y = t * sqrt(dxa^2 + dya^2) + point, z = t * dza + point, sizes = exp(t) * mean size, heading = atan2(sin, cos)
// The reference does this with ~25 elementwise torch kernels over (N, 8) tensors; same formulas, one read and one write.
namespace pdm {

__global__ __launch_bounds__(256) void point_head_decode_kernel(long long n, int num_class, const float *__restrict__ cls, int cls_stride,
                                                               const float *__restrict__ code, int code_stride,
                                                               const float *__restrict__ pts, int pts_stride,
                                                               const float *__restrict__ mean_size, float *__restrict__ boxes,
                                                               float *__restrict__ scores) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *c = cls + i * cls_stride;
    float best = c[0];
    int arg = 0;
    for (int k = 1; k < num_class; ++k) {
        const float v = c[k];
        if (v > best) { best = v; arg = k; }     // first maximum, as torch.max
    }
    scores[i] = 1.0f / (1.0f + expf(-best));
    const float4 t0 = *reinterpret_cast<const float4 *>(code + i * code_stride);
    const float4 t1 = *reinterpret_cast<const float4 *>(code + i * code_stride + 4);
    const float dxa = mean_size[arg * 3], dya = mean_size[arg * 3 + 1], dza = mean_size[arg * 3 + 2];
    const float diag = sqrtf(dxa * dxa + dya * dya);
    const float *p = pts + i * pts_stride;
    float *o = boxes + i * 7;
    o[0] = t0.x * diag + p[0];
    o[1] = t0.y * diag + p[1];
    o[2] = t0.z * dza + p[2];
    o[3] = expf(t0.w) * dxa;
    o[4] = expf(t1.x) * dya;
    o[5] = expf(t1.y) * dza;
    o[6] = atan2f(t1.w, t1.z);
}

}  // namespace pdm

// cls (N, cls_stride) logits with num_class live columns, code (N, code_stride >= 8) [x, y, z, dx, dy, dz, cos, sin] 16-byte
// aligned rows, pts (N, pts_stride >= 3), mean_size (num_class, 3) -> boxes (N, 7), scores (N).
extern "C" int pdm_point_head_decode(void *stream, long long n, int num_class, const float *cls, int cls_stride, const float *code,
                                     int code_stride, const float *pts, int pts_stride, const float *mean_size, float *boxes,
                                     float *scores) {
    using namespace pdm;
    PDM_REQUIRE(n >= 0 && num_class >= 1, PDM_E_BADARG, "point_head_decode: n=%lld num_class=%d", n, num_class);
    if (n == 0) return 0;
    PDM_REQUIRE(cls && code && pts && mean_size && boxes && scores, PDM_E_BADARG, "point_head_decode: null pointer");
    PDM_REQUIRE(cls_stride >= num_class && code_stride >= 8 && code_stride % 4 == 0 && pts_stride >= 3 &&
                    (reinterpret_cast<uintptr_t>(code) & 15) == 0, PDM_E_BADARG, "point_head_decode: strides / alignment");
    PDM_REQUIRE(n <= 0x7fffffffll * 256, PDM_E_TOOLARGE, "point_head_decode: too many points");
    hipLaunchKernelGGL(point_head_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), n, num_class, cls,
                       cls_stride, code, code_stride, pts, pts_stride, mean_size, boxes, scores);
    return check_launch("point_head_decode");
}
