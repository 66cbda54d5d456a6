// Training-mode BatchNorm + ReLU as one operator (forward and backward), for the shared MLPs of the train step
// (Conv/Linear -> BatchNorm -> ReLU triples of pointnet2_modules.py:19-55, point_head_template.py:35-48 of the
// reference, which run them as three torch modules).  Batch statistics in fp32 whatever the activation type (fp32 or
// bf16 under autocast), two layouts:
//   CL  rows x C, channel fastest   (Linear outputs, channels-last conv outputs: SA groups, BEV maps)
//   CF  n x C x L, position fastest (NCHW / NCL tensors: FP modules, the neck's projection)
// Passes: forward  = statistics (1 read) -> per-channel finalize -> normalise + ReLU (1 read, 1 write)
//         backward = two sums per channel (2 reads) -> finalize -> input gradient (2 reads, 1 write)
// Bound: HBM.  Partial sums go to a (parts, C, 2) buffer and are folded in double by the finalize kernels, so a run is
// bit-reproducible (no float atomics).
#include "common.h"

namespace pdm {

template <class T> struct BnVec;
template <> struct BnVec<float> {
    static constexpr int V = 4;
    typedef float4 raw;
    __device__ static void load(const float *p, float (&v)[4]) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
    __device__ static void store(float *p, const float (&v)[4]) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {   // round to nearest even; NaN stays NaN
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
template <> struct BnVec<bf16_t> {
    static constexpr int V = 8;
    __device__ static void load(const bf16_t *p, float (&v)[8]) {
        const uint4 q = *reinterpret_cast<const uint4 *>(p);
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    __device__ static void store(bf16_t *p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf(v[2 * i]) | ((unsigned)f2bf(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

struct BnCoef {   // per-channel constants, all (C) fp32
    const float *mean, *invstd, *scale, *shift;   // scale = gamma * invstd, shift = beta:  y = (x - mean) * scale + beta
    const float *p, *q;                           // backward: dx = scale * (g - p - (x - mean) * q), p = sum(g) / count, q = invstd sum(g xhat) / count
    float *pivot;                                 // statistics pass: per-channel shift of the sums (the channel's first element)
};

template <int V>
__device__ __forceinline__ void ldv(const float *p, float (&v)[V]) {   // V consecutive floats, 16-byte aligned
#pragma unroll
    for (int i = 0; i < V; i += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p + i);
        v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
}

template <class TD, int V>
__device__ __forceinline__ void ld4bf_any(const TD *p, float (&v)[V]) {   // four bf16 -> fp32 (the mixed form; V == 4 there)
    if constexpr (V == 4 && sizeof(TD) == 2) {
        const uint2 q = *reinterpret_cast<const uint2 *>(p);
        v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
        v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
    }
}

// ---- CL: rows x C.  A workgroup walks rows blockIdx.x * RPP + k * gridDim.x * RPP; thread = (row in pass, V channels) --
// MODE 0: sums of x and x^2.  MODE 1: sums of g and g * xhat with g = dy * [x scale + shift > 0] (or dy when !relu).
template <class T, int MODE, class TD = T>
__global__ __launch_bounds__(256) void bn_cl_reduce_kernel(const T *__restrict__ x, const TD *__restrict__ dy, long long rows, int C,
                                                           BnCoef k, int relu, float *__restrict__ partial) {
    constexpr int V = BnVec<T>::V;
    __shared__ float s_a[256 * V], s_b[256 * V];
    const int tpr = C / V, rpp = 256 / tpr;
    const int rin = threadIdx.x / tpr, col = (threadIdx.x % tpr) * V;
    float a[V], b[V], mean[V], istd[V], sc[V], sh[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { a[i] = b[i] = 0.f; mean[i] = istd[i] = sc[i] = sh[i] = 0.f; }
    if (rin < rpp) {
        if (MODE == 1) { ldv<V>(k.mean + col, mean); ldv<V>(k.invstd + col, istd); ldv<V>(k.scale + col, sc); ldv<V>(k.shift + col, sh); }
        else {
            // sums are taken of x - pivot (pivot = the channel's first element): E[x^2] - E[x]^2 then loses nothing to a large mean
            BnVec<T>::load(x + col, mean);
            if (blockIdx.x == 0 && rin == 0) {
#pragma unroll
                for (int i = 0; i < V; ++i) k.pivot[col + i] = mean[i];
            }
        }
    }
    if (rin < rpp) {
        for (long long r = (long long)blockIdx.x * rpp + rin; r < rows; r += (long long)gridDim.x * rpp) {
            float xv[V];
            BnVec<T>::load(x + r * C + col, xv);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < V; ++i) { const float d = xv[i] - mean[i]; a[i] += d; b[i] = fmaf(d, d, b[i]); }
            } else {
                float gv[V];
                if constexpr (sizeof(TD) == sizeof(T)) BnVec<T>::load(reinterpret_cast<const T *>(dy) + r * C + col, gv);
                else ld4bf_any(dy + r * C + col, gv);                      // mixed: fp32 x (V = 4), bf16 dy
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float d = xv[i] - mean[i];
                    const float g = (!relu || fmaf(d, sc[i], sh[i]) > 0.f) ? gv[i] : 0.f;
                    a[i] += g;
                    b[i] = fmaf(g, d * istd[i], b[i]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) { s_a[threadIdx.x * V + i] = a[i]; s_b[threadIdx.x * V + i] = b[i]; }
    __syncthreads();
    // channel c of this workgroup = sum over the rpp row slots; one thread per channel
    for (int c = threadIdx.x; c < C; c += 256) {
        float ta = 0.f, tb = 0.f;
        const int t0 = c / V, i = c % V;
        for (int q = 0; q < rpp; ++q) { ta += s_a[(q * tpr + t0) * V + i]; tb += s_b[(q * tpr + t0) * V + i]; }
        partial[((size_t)blockIdx.x * C + c) * 2] = ta;
        partial[((size_t)blockIdx.x * C + c) * 2 + 1] = tb;
    }
}

// MODE 0: y = [relu](x scale + shift).  MODE 1: dx = scale * (g - k1 - xhat k2).
template <class T, int MODE>
__global__ __launch_bounds__(256) void bn_cl_apply_kernel(const T *__restrict__ x, const T *__restrict__ dy, T *__restrict__ out,
                                                          long long nvec, int C, BnCoef k, int relu) {
    constexpr int V = BnVec<T>::V;
    const int tpr = C / V;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long long)gridDim.x * 256) {
        const int col = (int)(e % tpr) * V;
        float xv[V], ov[V], sc[V], sh[V], mu[V];
        BnVec<T>::load(x + e * V, xv);
        ldv<V>(k.scale + col, sc); ldv<V>(k.shift + col, sh); ldv<V>(k.mean + col, mu);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float y = fmaf(xv[i] - mu[i], sc[i], sh[i]);   // the centred form: no cancellation when |mean| >> std
                ov[i] = relu ? fmaxf(y, 0.f) : y;
            }
        } else {
            float gv[V], pp[V], qq[V];
            BnVec<T>::load(dy + e * V, gv);
            ldv<V>(k.p + col, pp); ldv<V>(k.q + col, qq);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float d = xv[i] - mu[i];
                const float g = (!relu || fmaf(d, sc[i], sh[i]) > 0.f) ? gv[i] : 0.f;
                ov[i] = sc[i] * (g - pp[i] - d * qq[i]);
            }
        }
        BnVec<T>::store(out + e * V, ov);
    }
}

// ---- MIXED (dtype 2, rows x C): x (the BatchNorm's input, e.g. the fp32 depthwise map) fp32, the activation side — y forward, dy
// backward — bf16, dx fp32; four channels per thread.  The arithmetic of the fp32 kernels with the conversion a bf16 consumer
// (a contraction under autocast) would do anyway folded into the store / the load: y = bf16(relu(bn(x))) is what
// `bn(x).relu().to(bf16)` gives, and a bf16 dy read as fp32 is exact — so results are bit for bit those of the fp32 operator
// followed / preceded by the casts, without a 577 MB cast pass each way on the heat-map head's map.
__device__ __forceinline__ void ld4bf(const bf16_t *p, float (&v)[4]) {
    const uint2 q = *reinterpret_cast<const uint2 *>(p);
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
    v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
}
template <int MODE>
__global__ __launch_bounds__(256) void bn_cl_apply_mixed_kernel(const float *__restrict__ x, const bf16_t *__restrict__ dy,
                                                                void *__restrict__ out, long long nvec, int C, BnCoef k, int relu) {
    const int tpr = C / 4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long long)gridDim.x * 256) {
        const int col = (int)(e % tpr) * 4;
        float xv[4], ov[4], sc[4], sh[4], mu[4];
        BnVec<float>::load(x + e * 4, xv);
        ldv<4>(k.scale + col, sc); ldv<4>(k.shift + col, sh); ldv<4>(k.mean + col, mu);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y = fmaf(xv[i] - mu[i], sc[i], sh[i]);
                ov[i] = relu ? fmaxf(y, 0.f) : y;
            }
            uint2 w;
            w.x = (unsigned)f2bf(ov[0]) | ((unsigned)f2bf(ov[1]) << 16);
            w.y = (unsigned)f2bf(ov[2]) | ((unsigned)f2bf(ov[3]) << 16);
            *reinterpret_cast<uint2 *>(static_cast<bf16_t *>(out) + e * 4) = w;
        } else {
            float gv[4], pp[4], qq[4];
            ld4bf(dy + e * 4, gv);
            ldv<4>(k.p + col, pp); ldv<4>(k.q + col, qq);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = xv[i] - mu[i];
                const float g = (!relu || fmaf(d, sc[i], sh[i]) > 0.f) ? gv[i] : 0.f;
                ov[i] = sc[i] * (g - pp[i] - d * qq[i]);
            }
            BnVec<float>::store(static_cast<float *>(out) + e * 4, ov);
        }
    }
}

// ---- CF: n x C x L.  blockIdx.x = channel, blockIdx.y = slice of the n * (L / V) vectors of that channel -------------
template <class T, int MODE>
__global__ __launch_bounds__(256) void bn_cf_reduce_kernel(const T *__restrict__ x, const T *__restrict__ dy, long long n, int C, long long L,
                                                           BnCoef k, int relu, float *__restrict__ partial) {
    constexpr int V = BnVec<T>::V;
    __shared__ float s_a[256], s_b[256];
    const int c = blockIdx.x;
    const long long lv = L / V, total = n * lv;
    float a = 0.f, b = 0.f;
    float mean = MODE ? k.mean[c] : 0.f;
    const float istd = MODE ? k.invstd[c] : 0.f, sc = MODE ? k.scale[c] : 0.f, sh = MODE ? k.shift[c] : 0.f;
    if (MODE == 0) {   // pivot = the channel's first element (see the CL kernel)
        float first[V];
        BnVec<T>::load(x + (size_t)c * L, first);
        mean = first[0];
        if (blockIdx.y == 0 && threadIdx.x == 0) k.pivot[c] = mean;
    }
    for (long long e = (long long)blockIdx.y * 256 + threadIdx.x; e < total; e += (long long)gridDim.y * 256) {
        const long long s = e / lv, l = (e - s * lv) * V;
        const size_t off = ((size_t)s * C + c) * L + l;
        float xv[V];
        BnVec<T>::load(x + off, xv);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < V; ++i) { const float d = xv[i] - mean; a += d; b = fmaf(d, d, b); }
        } else {
            float gv[V];
            BnVec<T>::load(dy + off, gv);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float d = xv[i] - mean;
                const float g = (!relu || fmaf(d, sc, sh) > 0.f) ? gv[i] : 0.f;
                a += g;
                b = fmaf(g, d * istd, b);
            }
        }
    }
    s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (threadIdx.x < half) { s_a[threadIdx.x] += s_a[threadIdx.x + half]; s_b[threadIdx.x] += s_b[threadIdx.x + half]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[((size_t)blockIdx.y * C + c) * 2] = s_a[0];
        partial[((size_t)blockIdx.y * C + c) * 2 + 1] = s_b[0];
    }
}

template <class T, int MODE>
__global__ __launch_bounds__(256) void bn_cf_apply_kernel(const T *__restrict__ x, const T *__restrict__ dy, T *__restrict__ out,
                                                          long long nvec, int C, long long L, BnCoef k, int relu) {
    constexpr int V = BnVec<T>::V;
    const long long lv = L / V;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long long)gridDim.x * 256) {
        const int c = (int)((e / lv) % C);
        float xv[V], ov[V];
        BnVec<T>::load(x + e * V, xv);
        const float sc = k.scale[c], sh = k.shift[c], mu = k.mean[c];
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float y = fmaf(xv[i] - mu, sc, sh);
                ov[i] = relu ? fmaxf(y, 0.f) : y;
            }
        } else {
            float gv[V];
            BnVec<T>::load(dy + e * V, gv);
            const float pp = k.p[c], qq = k.q[c];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float d = xv[i] - mu;
                const float g = (!relu || fmaf(d, sc, sh) > 0.f) ? gv[i] : 0.f;
                ov[i] = sc * (g - pp - d * qq);
            }
        }
        BnVec<T>::store(out + e * V, ov);
    }
}

// ---- BatchNorm + ReLU + max over the `ns` neighbours of a group (the tail of an SA scale: pointnet2_modules.py:46-52) ----
// x is (G groups, ns, C) with the channel fastest.  relu((x - mean) scale + beta) is monotone in x (rising for scale >= 0,
// falling otherwise), so the pooled output is that function of the group's max (or min) of x: ONE pass over x yields
// the batch statistics AND every group's max / min with the first index attaining them; the normalised tensor is never
// written.  Backward: the pooled gradient belongs to that one element, so the two sums run over the pooled tensors
// only, and the input gradient is one pass (read x, write dx).  Thread = (group, V channels), looping over s.
template <class T>
__global__ __launch_bounds__(256) void bn_pool_stats_kernel(const T *__restrict__ x, long long G, int ns, int C, float *pivot,
                                                            T *__restrict__ xmax, T *__restrict__ xmin,
                                                            unsigned char *__restrict__ imax, unsigned char *__restrict__ imin,
                                                            float *__restrict__ partial) {
    constexpr int V = BnVec<T>::V;
    __shared__ float s_a[256 * V], s_b[256 * V];
    const int tpr = C / V, gpp = 256 / tpr;
    const int gin = threadIdx.x / tpr, col = (threadIdx.x % tpr) * V;
    float a[V], b[V], pv[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { a[i] = b[i] = 0.f; pv[i] = 0.f; }
    if (gin < gpp) {
        BnVec<T>::load(x + col, pv);
        if (blockIdx.x == 0 && gin == 0) {
#pragma unroll
            for (int i = 0; i < V; ++i) pivot[col + i] = pv[i];
        }
        for (long long g = (long long)blockIdx.x * gpp + gin; g < G; g += (long long)gridDim.x * gpp) {
            float mx[V], mn[V];
            int ix[V], in_[V];
#pragma unroll
            for (int i = 0; i < V; ++i) { mx[i] = -INFINITY; mn[i] = INFINITY; ix[i] = in_[i] = 0; }
            const T *__restrict__ row = x + (size_t)g * ns * C + col;
            for (int s_ = 0; s_ < ns; ++s_) {
                float xv[V];
                BnVec<T>::load(row + (size_t)s_ * C, xv);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float d = xv[i] - pv[i];
                    a[i] += d; b[i] = fmaf(d, d, b[i]);
                    if (xv[i] > mx[i]) { mx[i] = xv[i]; ix[i] = s_; }   // strict: the FIRST index attaining the extreme
                    if (xv[i] < mn[i]) { mn[i] = xv[i]; in_[i] = s_; }
                }
            }
            BnVec<T>::store(xmax + (size_t)g * C + col, mx);
            BnVec<T>::store(xmin + (size_t)g * C + col, mn);
#pragma unroll
            for (int i = 0; i < V; ++i) { imax[(size_t)g * C + col + i] = (unsigned char)ix[i]; imin[(size_t)g * C + col + i] = (unsigned char)in_[i]; }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) { s_a[threadIdx.x * V + i] = a[i]; s_b[threadIdx.x * V + i] = b[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float ta = 0.f, tb = 0.f;
        const int t0 = c / V, i = c % V;
        for (int q = 0; q < gpp; ++q) { ta += s_a[(q * tpr + t0) * V + i]; tb += s_b[(q * tpr + t0) * V + i]; }
        partial[((size_t)blockIdx.x * C + c) * 2] = ta;
        partial[((size_t)blockIdx.x * C + c) * 2 + 1] = tb;
    }
}

// MODE 0: y[g][c] = [relu]((sel - mean) scale + beta), sel = scale >= 0 ? xmax : xmin.
// MODE 1: the backward's two sums over the pooled tensors (g = dy where the pooled value passed the ReLU).
template <class T, int MODE>
__global__ __launch_bounds__(256) void bn_pool_small_kernel(const T *__restrict__ xmax, const T *__restrict__ xmin, const T *__restrict__ dy,
                                                            T *__restrict__ y, long long G, int C, BnCoef k, int relu,
                                                            float *__restrict__ partial) {
    constexpr int V = BnVec<T>::V;
    __shared__ float s_a[MODE ? 256 * V : 1], s_b[MODE ? 256 * V : 1];
    const int tpr = C / V, rpp = 256 / tpr;
    const int rin = threadIdx.x / tpr, col = (threadIdx.x % tpr) * V;
    float a[V], b[V], mean[V], istd[V], sc[V], sh[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { a[i] = b[i] = 0.f; mean[i] = istd[i] = sc[i] = sh[i] = 0.f; }
    if (rin < rpp) {
        ldv<V>(k.mean + col, mean); ldv<V>(k.invstd + col, istd); ldv<V>(k.scale + col, sc); ldv<V>(k.shift + col, sh);
        for (long long g = (long long)blockIdx.x * rpp + rin; g < G; g += (long long)gridDim.x * rpp) {
            float hi[V], lo[V];
            BnVec<T>::load(xmax + (size_t)g * C + col, hi);
            BnVec<T>::load(xmin + (size_t)g * C + col, lo);
            if (MODE == 0) {
                float ov[V];
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float v = fmaf((sc[i] >= 0.f ? hi[i] : lo[i]) - mean[i], sc[i], sh[i]);
                    ov[i] = relu ? fmaxf(v, 0.f) : v;
                }
                BnVec<T>::store(y + (size_t)g * C + col, ov);
            } else {
                float gv[V];
                BnVec<T>::load(dy + (size_t)g * C + col, gv);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float d = (sc[i] >= 0.f ? hi[i] : lo[i]) - mean[i];
                    const float gg = (!relu || fmaf(d, sc[i], sh[i]) > 0.f) ? gv[i] : 0.f;
                    a[i] += gg;
                    b[i] = fmaf(gg, d * istd[i], b[i]);
                }
            }
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < V; ++i) { s_a[threadIdx.x * V + i] = a[i]; s_b[threadIdx.x * V + i] = b[i]; }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float ta = 0.f, tb = 0.f;
            const int t0 = c / V, i = c % V;
            for (int q = 0; q < rpp; ++q) { ta += s_a[(q * tpr + t0) * V + i]; tb += s_b[(q * tpr + t0) * V + i]; }
            partial[((size_t)blockIdx.x * C + c) * 2] = ta;
            partial[((size_t)blockIdx.x * C + c) * 2 + 1] = tb;
        }
    }
}

// dx[g][s][c] = scale * (g_sel - p - (x - mean) q), g_sel = dy[g][c] at the selected s (if it passed the ReLU), else 0
template <class T>
__global__ __launch_bounds__(256) void bn_pool_dx_kernel(const T *__restrict__ x, const T *__restrict__ dy, const T *__restrict__ xmax,
                                                         const T *__restrict__ xmin, const unsigned char *__restrict__ imax,
                                                         const unsigned char *__restrict__ imin, T *__restrict__ dx, long long G, int ns,
                                                         int C, BnCoef k, int relu) {
    constexpr int V = BnVec<T>::V;
    const int tpr = C / V, gpp = 256 / tpr;
    const int gin = threadIdx.x / tpr, col = (threadIdx.x % tpr) * V;
    if (gin >= gpp) return;
    float mean[V], sc[V], sh[V], pp[V], qq[V];
    ldv<V>(k.mean + col, mean); ldv<V>(k.scale + col, sc); ldv<V>(k.shift + col, sh); ldv<V>(k.p + col, pp); ldv<V>(k.q + col, qq);
    for (long long g = (long long)blockIdx.x * gpp + gin; g < G; g += (long long)gridDim.x * gpp) {
        float hi[V], lo[V], gv[V];
        int sel[V];
        BnVec<T>::load(xmax + (size_t)g * C + col, hi);
        BnVec<T>::load(xmin + (size_t)g * C + col, lo);
        BnVec<T>::load(dy + (size_t)g * C + col, gv);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const bool up = sc[i] >= 0.f;
            const float d = (up ? hi[i] : lo[i]) - mean[i];
            if (relu && !(fmaf(d, sc[i], sh[i]) > 0.f)) gv[i] = 0.f;
            sel[i] = up ? imax[(size_t)g * C + col + i] : imin[(size_t)g * C + col + i];
        }
        const T *__restrict__ row = x + (size_t)g * ns * C + col;
        T *__restrict__ orow = dx + (size_t)g * ns * C + col;
        for (int s_ = 0; s_ < ns; ++s_) {
            float xv[V], ov[V];
            BnVec<T>::load(row + (size_t)s_ * C, xv);
#pragma unroll
            for (int i = 0; i < V; ++i) ov[i] = sc[i] * ((s_ == sel[i] ? gv[i] : 0.f) - pp[i] - (xv[i] - mean[i]) * qq[i]);
            BnVec<T>::store(orow + (size_t)s_ * C, ov);
        }
    }
}

// ---- per-channel finalize: one wave per channel folds the parts in double -------------------------------------------
// (four 8-byte loads in flight per lane, a butterfly of wave shuffles instead of six barriers over LDS: every lane ends with the
// sums, added in a fixed order)
__device__ __forceinline__ void bn_fold(const float *__restrict__ partial, int parts, int C, int c, double &s, double &q) {
    const float2 *__restrict__ src = reinterpret_cast<const float2 *>(partial) + c;
    double ls = 0.0, lq = 0.0;
    int p = threadIdx.x;
    for (; p + 192 < parts; p += 256) {
        const float2 v0 = src[(size_t)p * C], v1 = src[(size_t)(p + 64) * C], v2 = src[(size_t)(p + 128) * C], v3 = src[(size_t)(p + 192) * C];
        ls += v0.x; lq += v0.y; ls += v1.x; lq += v1.y; ls += v2.x; lq += v2.y; ls += v3.x; lq += v3.y;
    }
    for (; p < parts; p += 64) { const float2 v = src[(size_t)p * C]; ls += v.x; lq += v.y; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { ls += __shfl_xor(ls, off, 64); lq += __shfl_xor(lq, off, 64); }
    s = ls; q = lq;
}
// forward: mean, biased variance -> invstd, scale, shift; running statistics updated as torch.nn.BatchNorm does
// (momentum, unbiased variance).  coef layout: [mean | invstd | gamma invstd | beta] (4, C); on entry row 3 holds the pivots.
__global__ __launch_bounds__(64) void bn_finalize_fwd_kernel(const float *__restrict__ partial, int parts, int C, double count,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                             float momentum, float *running_mean, float *running_var,
                                                             float *__restrict__ coef, int zero_pivot = 0) {
    const int c = blockIdx.x;
    double s, q;
    bn_fold(partial, parts, C, c, s, q);
    if (threadIdx.x != 0) return;
    const double dm = s / count, mean = (zero_pivot ? 0.0 : (double)coef[3 * C + c]) + dm;   // zero_pivot: plain sums (no parked pivot row)
    double var = q / count - dm * dm;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float scale = g * invstd;
    coef[c] = (float)mean; coef[C + c] = invstd; coef[2 * C + c] = scale; coef[3 * C + c] = b;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(count > 1.0 ? var * count / (count - 1.0) : var);
}
// backward: dbeta = sum g, dgamma = sum g xhat; the input gradient is dx = scale (g - p - (x - mean) q) with
// p = dbeta / count, q = invstd dgamma / count.  out layout [dgamma | dbeta | p | q]
__global__ __launch_bounds__(64) void bn_finalize_bwd_kernel(const float *__restrict__ partial, int parts, int C, double count,
                                                             const float *__restrict__ coef, float *__restrict__ out) {
    const int c = blockIdx.x;
    double s, q;
    bn_fold(partial, parts, C, c, s, q);
    if (threadIdx.x != 0) return;
    out[c] = (float)q; out[C + c] = (float)s;
    out[2 * C + c] = (float)(s / count);
    out[3 * C + c] = (float)((double)coef[C + c] * q / count);
}

static int bn_check(const char *who, int dtype, int layout, long long n, int C, long long L, const void *p0, const void *p1) {
    PDM_REQUIRE(dtype == 0 || dtype == 1 || (dtype == 2 && layout == 0), PDM_E_BADARG,
                "%s: dtype %d (0 = fp32, 1 = bf16, 2 = fp32 x with bf16 y / dy, rows x C only)", who, dtype);
    PDM_REQUIRE(layout == 0 || layout == 1, PDM_E_BADARG, "%s: layout %d (0 = rows x C, 1 = n x C x L)", who, layout);
    const int V = dtype == 1 ? 8 : 4;
    PDM_REQUIRE(n >= 0 && C >= 1 && L >= 1, PDM_E_BADARG, "%s: n=%lld C=%d L=%lld", who, n, C, L);
    if (layout == 0) PDM_REQUIRE(C % V == 0 && C / V <= 256, PDM_E_BADARG, "%s: rows x C needs C a multiple of %d, at most %d", who, V, 256 * V);
    else PDM_REQUIRE(L % V == 0 && C <= 65535, PDM_E_BADARG, "%s: n x C x L needs L a multiple of %d", who, V);
    PDM_REQUIRE(n == 0 || (p0 && p1), PDM_E_BADARG, "%s: null pointer", who);
    PDM_REQUIRE(((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1)) & 15) == 0, PDM_E_BADARG, "%s: buffers must be 16-byte aligned", who);
    return 0;
}

static BnCoef coef_of(const float *coef, const float *bwd, int C) {
    BnCoef k{};
    if (coef) { k.mean = coef; k.invstd = coef + C; k.scale = coef + 2 * C; k.shift = coef + 3 * C; }
    if (bwd) { k.p = bwd + 2 * C; k.q = bwd + 3 * C; }
    return k;
}

}  // namespace pdm

using namespace pdm;

// Number of partial-sum slices the reduce passes write for this shape ((parts, C, 2) floats of workspace).
extern "C" int pdm_bn_parts(int layout, long long n, int C, long long L) {
    const long long elems = n * (layout ? L : 1) * (layout ? 1 : C);
    if (layout == 0) { const long long want = (elems + 65535) / 65536; return (int)(want < 1 ? 1 : want > 1024 ? 1024 : want); }
    const long long want = (n * L + 32767) / 32768;   // slices per channel
    return (int)(want < 1 ? 1 : want > 64 ? 64 : want);
}

// Forward: statistics + finalize + y = [relu]((x - mean) invstd gamma + beta).  coef (4, C) fp32 is kept for the backward.
// n = rows (layout 0, L ignored) or batch entries (layout 1).  running_mean / running_var may be null.
extern "C" int pdm_bn_relu_forward(void *stream, int dtype, int layout, long long n, int C, long long L, const void *x, void *y,
                                   const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                   float *running_var, float *coef, float *partial, int relu) {
    if (layout == 0) L = 1;
    if (int rc = bn_check("bn_relu_forward", dtype, layout, n, C, L, x, y)) return rc;
    PDM_REQUIRE(coef && partial, PDM_E_BADARG, "bn_relu_forward: null workspace");
    if (n == 0) return 0;
    const int parts = pdm_bn_parts(layout, n, C, L);
    BnCoef none{};
    none.pivot = coef + 3 * (size_t)C;   // parked in the shift row until the finalize kernel replaces it
    const int V = dtype == 1 ? 8 : 4;
    const long long nvec = n * C * L / V;
    const long long ag = (nvec + 255) / 256;
    const dim3 agrid((unsigned)(ag > 16384 ? 16384 : ag));
    if (layout == 0) {
        if (dtype == 1) hipLaunchKernelGGL((bn_cl_reduce_kernel<bf16_t, 0>), dim3(parts), dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)nullptr, n, C, none, relu, partial);
        else hipLaunchKernelGGL((bn_cl_reduce_kernel<float, 0>), dim3(parts), dim3(256), 0, as_stream(stream), (const float *)x, (const float *)nullptr, n, C, none, relu, partial);
    } else {
        if (dtype) hipLaunchKernelGGL((bn_cf_reduce_kernel<bf16_t, 0>), dim3(C, parts), dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)nullptr, n, C, L, none, relu, partial);
        else hipLaunchKernelGGL((bn_cf_reduce_kernel<float, 0>), dim3(C, parts), dim3(256), 0, as_stream(stream), (const float *)x, (const float *)nullptr, n, C, L, none, relu, partial);
    }
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(C), dim3(64), 0, as_stream(stream), partial, parts, C, (double)n * (double)L, gamma,
                       beta, eps, momentum, running_mean, running_var, coef);
    const BnCoef k = coef_of(coef, nullptr, C);
    if (layout == 0) {
        if (dtype == 2) hipLaunchKernelGGL((bn_cl_apply_mixed_kernel<0>), agrid, dim3(256), 0, as_stream(stream), (const float *)x, (const bf16_t *)nullptr, y, nvec, C, k, relu);
        else if (dtype) hipLaunchKernelGGL((bn_cl_apply_kernel<bf16_t, 0>), agrid, dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)nullptr, (bf16_t *)y, nvec, C, k, relu);
        else hipLaunchKernelGGL((bn_cl_apply_kernel<float, 0>), agrid, dim3(256), 0, as_stream(stream), (const float *)x, (const float *)nullptr, (float *)y, nvec, C, k, relu);
    } else {
        if (dtype) hipLaunchKernelGGL((bn_cf_apply_kernel<bf16_t, 0>), agrid, dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)nullptr, (bf16_t *)y, nvec, C, L, k, relu);
        else hipLaunchKernelGGL((bn_cf_apply_kernel<float, 0>), agrid, dim3(256), 0, as_stream(stream), (const float *)x, (const float *)nullptr, (float *)y, nvec, C, L, k, relu);
    }
    return check_launch("bn_relu_forward");
}

// The same forward when the statistics pass has already been done by the producer of x: pdm_tg_gemm_nt (train_gemm.hip)
// leaves per-row-tile column sums of y and y^2 of its ROUNDED outputs, exactly the two sums bn_cl_reduce_kernel<MODE 0> forms
// (with pivot 0) — `partial` = [parts][C][2].  One pass over x (the apply) instead of two.  rows x C layout only.
extern "C" int pdm_bn_relu_forward_stats(void *stream, int dtype, long long n, int C, const void *x, void *y, const float *gamma,
                                         const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                         float *coef, const float *partial, int parts, int relu) {
    if (int rc = bn_check("bn_relu_forward_stats", dtype, 0, n, C, 1, x, y)) return rc;
    PDM_REQUIRE(coef && partial && parts >= 1, PDM_E_BADARG, "bn_relu_forward_stats: null workspace");
    if (n == 0) return 0;
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(C), dim3(64), 0, as_stream(stream), partial, parts, C, (double)n, gamma, beta, eps,
                       momentum, running_mean, running_var, coef, 1);   // the producer's sums are plain sums: pivot 0
    const BnCoef k = coef_of(coef, nullptr, C);
    const int V = dtype == 1 ? 8 : 4;     // dtype 2 = fp32 x, bf16 y: four elements per thread like the fp32 form
    const long long nvec = n * C / V;
    const long long ag = (nvec + 255) / 256;
    const dim3 agrid((unsigned)(ag > 16384 ? 16384 : ag));
    if (dtype == 2) hipLaunchKernelGGL((bn_cl_apply_mixed_kernel<0>), agrid, dim3(256), 0, as_stream(stream), (const float *)x, (const bf16_t *)nullptr, y, nvec, C, k, relu);
    else if (dtype) hipLaunchKernelGGL((bn_cl_apply_kernel<bf16_t, 0>), agrid, dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)nullptr, (bf16_t *)y, nvec, C, k, relu);
    else hipLaunchKernelGGL((bn_cl_apply_kernel<float, 0>), agrid, dim3(256), 0, as_stream(stream), (const float *)x, (const float *)nullptr, (float *)y, nvec, C, k, relu);
    return check_launch("bn_relu_forward_stats");
}

// The statistics half of pdm_bn_relu_forward on its own (rows x C): reduce + finalize -> coef (4, C), running statistics updated, and NO
// normalised tensor — for a consumer that applies BatchNorm + ReLU while it reads x (pdm_tg_gemm_nt / pdm_tg_wgrad with x_bn_coef) when
// x does not come from a contraction that could have taken the sums itself.  partial: pdm_bn_parts(0, n, C, 1) * C * 2 floats.
extern "C" int pdm_bn_forward_coef(void *stream, int dtype, long long n, int C, const void *x, const float *gamma, const float *beta, float eps,
                                   float momentum, float *running_mean, float *running_var, float *coef, float *partial) {
    if (int rc = bn_check("bn_forward_coef", dtype == 2 ? 0 : dtype, 0, n, C, 1, x, x)) return rc;
    PDM_REQUIRE(dtype == 0 || dtype == 1, PDM_E_BADARG, "bn_forward_coef: dtype %d (0 = fp32, 1 = bf16)", dtype);
    PDM_REQUIRE(coef && partial && n >= 1, PDM_E_BADARG, "bn_forward_coef: null workspace or no rows");
    const int parts = pdm_bn_parts(0, n, C, 1);
    BnCoef none{};
    none.pivot = coef + 3 * (size_t)C;   // parked in the shift row until the finalize kernel replaces it
    if (dtype == 1) hipLaunchKernelGGL((bn_cl_reduce_kernel<bf16_t, 0>), dim3(parts), dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)nullptr, n, C, none, 1, partial);
    else hipLaunchKernelGGL((bn_cl_reduce_kernel<float, 0>), dim3(parts), dim3(256), 0, as_stream(stream), (const float *)x, (const float *)nullptr, n, C, none, 1, partial);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(C), dim3(64), 0, as_stream(stream), partial, parts, C, (double)n, gamma, beta, eps, momentum,
                       running_mean, running_var, coef);
    return check_launch("bn_forward_coef");
}

// Only the finalize step of the above: coef (4, C) from the producer's sums, running statistics updated.  For a consumer that
// applies the normalisation itself while reading x (pdm_tg_gemm_nt / pdm_tg_wgrad with x_bn_coef).
extern "C" int pdm_bn_finalize_stats(void *stream, long long n, int C, const float *gamma, const float *beta, float eps, float momentum,
                                     float *running_mean, float *running_var, float *coef, const float *partial, int parts) {
    PDM_REQUIRE(n >= 1 && C >= 1 && coef && partial && parts >= 1, PDM_E_BADARG, "bn_finalize_stats: bad argument");
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(C), dim3(64), 0, as_stream(stream), partial, parts, C, (double)n, gamma, beta, eps,
                       momentum, running_mean, running_var, coef, 1);
    return check_launch("bn_finalize_stats");
}

// Backward: dx, and grads (4, C) = [dgamma | dbeta | k1 | k2] (the caller reads the first two rows).
// phase 1 = the statistics of the gradient (reduce + finalize -> grads), phase 2 = dx from x, dy, coef and grads; 3 = both.
static int bn_relu_backward_phases(const char *who, int phase, void *stream, int dtype, int layout, long long n, int C, long long L,
                                   const void *x, const void *dy, void *dx, const float *coef, float *grads, float *partial, int relu) {
    if (layout == 0) L = 1;
    if (int rc = bn_check(who, dtype, layout, n, C, L, x, dy)) return rc;
    PDM_REQUIRE(coef && grads && ((phase & 1) == 0 || partial) && ((phase & 2) == 0 || n == 0 || dx), PDM_E_BADARG, "%s: null pointer", who);
    PDM_REQUIRE((reinterpret_cast<uintptr_t>(dx) & 15) == 0, PDM_E_BADARG, "%s: dx must be 16-byte aligned", who);
    if (n == 0) return 0;
    const int parts = pdm_bn_parts(layout, n, C, L);
    const int V = dtype == 1 ? 8 : 4;
    const long long nvec = n * C * L / V;
    const long long ag = (nvec + 255) / 256;
    const dim3 agrid((unsigned)(ag > 16384 ? 16384 : ag));
    BnCoef k = coef_of(coef, nullptr, C);
    if (phase & 1) {
        if (layout == 0) {
            if (dtype == 2) hipLaunchKernelGGL((bn_cl_reduce_kernel<float, 1, bf16_t>), dim3(parts), dim3(256), 0, as_stream(stream), (const float *)x, (const bf16_t *)dy, n, C, k, relu, partial);
            else if (dtype) hipLaunchKernelGGL((bn_cl_reduce_kernel<bf16_t, 1>), dim3(parts), dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)dy, n, C, k, relu, partial);
            else hipLaunchKernelGGL((bn_cl_reduce_kernel<float, 1>), dim3(parts), dim3(256), 0, as_stream(stream), (const float *)x, (const float *)dy, n, C, k, relu, partial);
        } else {
            if (dtype) hipLaunchKernelGGL((bn_cf_reduce_kernel<bf16_t, 1>), dim3(C, parts), dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)dy, n, C, L, k, relu, partial);
            else hipLaunchKernelGGL((bn_cf_reduce_kernel<float, 1>), dim3(C, parts), dim3(256), 0, as_stream(stream), (const float *)x, (const float *)dy, n, C, L, k, relu, partial);
        }
        hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(C), dim3(64), 0, as_stream(stream), partial, parts, C, (double)n * (double)L, coef, grads);
    }
    if (phase & 2) {
        k = coef_of(coef, grads, C);
        if (layout == 0) {
            if (dtype == 2) hipLaunchKernelGGL((bn_cl_apply_mixed_kernel<1>), agrid, dim3(256), 0, as_stream(stream), (const float *)x, (const bf16_t *)dy, dx, nvec, C, k, relu);
            else if (dtype) hipLaunchKernelGGL((bn_cl_apply_kernel<bf16_t, 1>), agrid, dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)dy, (bf16_t *)dx, nvec, C, k, relu);
            else hipLaunchKernelGGL((bn_cl_apply_kernel<float, 1>), agrid, dim3(256), 0, as_stream(stream), (const float *)x, (const float *)dy, (float *)dx, nvec, C, k, relu);
        } else {
            if (dtype) hipLaunchKernelGGL((bn_cf_apply_kernel<bf16_t, 1>), agrid, dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)dy, (bf16_t *)dx, nvec, C, L, k, relu);
            else hipLaunchKernelGGL((bn_cf_apply_kernel<float, 1>), agrid, dim3(256), 0, as_stream(stream), (const float *)x, (const float *)dy, (float *)dx, nvec, C, L, k, relu);
        }
    }
    return check_launch(who);
}

extern "C" int pdm_bn_relu_backward(void *stream, int dtype, int layout, long long n, int C, long long L, const void *x, const void *dy,
                                    void *dx, const float *coef, float *grads, float *partial, int relu) {
    return bn_relu_backward_phases("bn_relu_backward", 3, stream, dtype, layout, n, C, L, x, dy, dx, coef, grads, partial, relu);
}

// The two halves of pdm_bn_relu_backward on their own: `_stats` leaves grads (4, C) = [dgamma | dbeta | p | q] and writes no dx —
// for a consumer that forms dx while it reads dy and x (pdm_tg_gemm_nt_dy); `_apply` writes dx from grads already there.
extern "C" int pdm_bn_relu_backward_stats(void *stream, int dtype, int layout, long long n, int C, long long L, const void *x,
                                          const void *dy, const float *coef, float *grads, float *partial, int relu) {
    return bn_relu_backward_phases("bn_relu_backward_stats", 1, stream, dtype, layout, n, C, L, x, dy, nullptr, coef, grads, partial, relu);
}
extern "C" int pdm_bn_relu_backward_apply(void *stream, int dtype, int layout, long long n, int C, long long L, const void *x,
                                          const void *dy, void *dx, const float *coef, float *grads, int relu) {
    return bn_relu_backward_phases("bn_relu_backward_apply", 2, stream, dtype, layout, n, C, L, x, dy, dx, coef, grads, nullptr, relu);
}

// Only the finalize step of pdm_bn_relu_backward_stats: grads (4, C) = [dgamma | dbeta | p | q] from sums a producer has already taken —
// pdm_tg_gemm_nt_bs / pdm_tg_gemm_nt_dy_bs leave [parts][C][2] = per slot sum g and sum g xhat of the gradient they produce
// (train_gemm.hip); folded in double, in slot order.
extern "C" int pdm_bn_finalize_bwd_stats(void *stream, long long n, int C, const float *coef, float *grads, const float *partial, int parts) {
    PDM_REQUIRE(n >= 1 && C >= 1 && coef && grads && partial && parts >= 1, PDM_E_BADARG, "bn_finalize_bwd_stats: bad argument");
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(C), dim3(64), 0, as_stream(stream), partial, parts, C, (double)n, coef, grads);
    return check_launch("bn_finalize_bwd_stats");
}

static int bn_pool_check(const char *who, int dtype, long long G, int ns, int C, const void *a, const void *b, const void *c,
                         const void *d, const void *e, const void *f) {
    PDM_REQUIRE(dtype == 0 || dtype == 1, PDM_E_BADARG, "%s: dtype %d (0 = fp32, 1 = bf16)", who, dtype);
    const int V = dtype ? 8 : 4;
    PDM_REQUIRE(G >= 0 && ns >= 1 && ns <= 255 && C >= 1 && C % V == 0 && C / V <= 256, PDM_E_BADARG,
                "%s: G=%lld ns=%d (1..255) C=%d (a multiple of %d, at most %d)", who, G, ns, C, V, 256 * V);
    PDM_REQUIRE(G == 0 || (a && b && c && d && e && f), PDM_E_BADARG, "%s: null pointer", who);
    PDM_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                  reinterpret_cast<uintptr_t>(d)) & 15) == 0, PDM_E_BADARG, "%s: buffers must be 16-byte aligned", who);
    return 0;
}

static unsigned bn_pool_grid(long long units, int per_block) {
    const long long want = (units + per_block - 1) / per_block;
    return (unsigned)(want < 1 ? 1 : want > 1024 ? 1024 : want);
}

// Number of partial-sum slices of the pooled operator's passes ((parts, C, 2) floats of workspace).
extern "C" int pdm_bn_pool_parts(int dtype, long long G, int C) {
    const int V = dtype ? 8 : 4;
    if (C < V || C / V > 256) return 1;
    return (int)bn_pool_grid(G, 256 / (C / V));
}

// BatchNorm(train) + ReLU + max over the ns rows of each group, x (G, ns, C) channel fastest -> y (G, C).
// xmax / xmin (G, C) of x's type and imax / imin (G, C) bytes are kept for the backward, with coef (4, C).
extern "C" int pdm_bn_relu_pool_forward(void *stream, int dtype, long long G, int ns, int C, const void *x, void *y, void *xmax,
                                        void *xmin, unsigned char *imax, unsigned char *imin, const float *gamma,
                                        const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                        float *coef, float *partial, int relu) {
    if (int rc = bn_pool_check("bn_relu_pool_forward", dtype, G, ns, C, x, y, xmax, xmin, imax, imin)) return rc;
    PDM_REQUIRE(coef && partial, PDM_E_BADARG, "bn_relu_pool_forward: null workspace");
    if (G == 0) return 0;
    const int V = dtype ? 8 : 4, per = 256 / (C / V);
    const unsigned parts = bn_pool_grid(G, per);
    float *pivot = coef + 3 * (size_t)C;
    if (dtype) hipLaunchKernelGGL((bn_pool_stats_kernel<bf16_t>), dim3(parts), dim3(256), 0, as_stream(stream), (const bf16_t *)x, G, ns, C, pivot, (bf16_t *)xmax, (bf16_t *)xmin, imax, imin, partial);
    else hipLaunchKernelGGL((bn_pool_stats_kernel<float>), dim3(parts), dim3(256), 0, as_stream(stream), (const float *)x, G, ns, C, pivot, (float *)xmax, (float *)xmin, imax, imin, partial);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(C), dim3(64), 0, as_stream(stream), partial, (int)parts, C, (double)G * (double)ns, gamma, beta,
                       eps, momentum, running_mean, running_var, coef);
    const BnCoef k = coef_of(coef, nullptr, C);
    if (dtype) hipLaunchKernelGGL((bn_pool_small_kernel<bf16_t, 0>), dim3(parts), dim3(256), 0, as_stream(stream), (const bf16_t *)xmax, (const bf16_t *)xmin, (const bf16_t *)nullptr, (bf16_t *)y, G, C, k, relu, (float *)nullptr);
    else hipLaunchKernelGGL((bn_pool_small_kernel<float, 0>), dim3(parts), dim3(256), 0, as_stream(stream), (const float *)xmax, (const float *)xmin, (const float *)nullptr, (float *)y, G, C, k, relu, (float *)nullptr);
    return check_launch("bn_relu_pool_forward");
}

// The same forward when the producer of x has already taken everything the statistics pass computes: pdm_tg_gemm_nt_pool
// (train_gemm.hip) leaves the column sums of x and x^2 per slot (`partial` = [parts][C][2], plain sums) and every group's max / min
// with their first indices.  What is left: the finalize (coef, running statistics) and the pooled output.  bf16 only (dtype 1).
extern "C" int pdm_bn_relu_pool_forward_kept(void *stream, int dtype, long long G, int ns, int C, void *y, const void *xmax, const void *xmin,
                                             const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                             float *running_var, float *coef, const float *partial, int parts, int relu) {
    PDM_REQUIRE(dtype == 1, PDM_E_BADARG, "bn_relu_pool_forward_kept: dtype %d (1 = bf16)", dtype);
    PDM_REQUIRE(G >= 0 && ns >= 1 && ns <= 255 && C >= 8 && C % 8 == 0 && C / 8 <= 256, PDM_E_BADARG,
                "bn_relu_pool_forward_kept: G=%lld ns=%d C=%d", G, ns, C);
    if (G == 0) return 0;
    PDM_REQUIRE(y && xmax && xmin && coef && partial && parts >= 1, PDM_E_BADARG, "bn_relu_pool_forward_kept: null pointer");
    PDM_REQUIRE(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(xmax) | reinterpret_cast<uintptr_t>(xmin)) & 15) == 0, PDM_E_BADARG,
                "bn_relu_pool_forward_kept: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(C), dim3(64), 0, as_stream(stream), partial, parts, C, (double)G * (double)ns, gamma, beta,
                       eps, momentum, running_mean, running_var, coef, 1);
    const int per = 256 / (C / 8);
    const unsigned grid = bn_pool_grid(G, per);
    const BnCoef k = coef_of(coef, nullptr, C);
    hipLaunchKernelGGL((bn_pool_small_kernel<bf16_t, 0>), dim3(grid), dim3(256), 0, as_stream(stream), (const bf16_t *)xmax, (const bf16_t *)xmin,
                       (const bf16_t *)nullptr, (bf16_t *)y, G, C, k, relu, (float *)nullptr);
    return check_launch("bn_relu_pool_forward_kept");
}

// Backward of the above: dx (G, ns, C) from dy (G, C); grads (4, C) = [dgamma | dbeta | p | q].
extern "C" int pdm_bn_relu_pool_backward(void *stream, int dtype, long long G, int ns, int C, const void *x, const void *dy, void *dx,
                                         const void *xmax, const void *xmin, const unsigned char *imax, const unsigned char *imin,
                                         const float *coef, float *grads, float *partial, int relu) {
    if (int rc = bn_pool_check("bn_relu_pool_backward", dtype, G, ns, C, x, dy, xmax, xmin, imax, imin)) return rc;
    PDM_REQUIRE(coef && grads && partial && (G == 0 || dx) && (reinterpret_cast<uintptr_t>(dx) & 15) == 0, PDM_E_BADARG,
                "bn_relu_pool_backward: null or unaligned pointer");
    if (G == 0) return 0;
    const int V = dtype ? 8 : 4, per = 256 / (C / V);
    const unsigned parts = bn_pool_grid(G, per);
    BnCoef k = coef_of(coef, nullptr, C);
    if (dtype) hipLaunchKernelGGL((bn_pool_small_kernel<bf16_t, 1>), dim3(parts), dim3(256), 0, as_stream(stream), (const bf16_t *)xmax, (const bf16_t *)xmin, (const bf16_t *)dy, (bf16_t *)nullptr, G, C, k, relu, partial);
    else hipLaunchKernelGGL((bn_pool_small_kernel<float, 1>), dim3(parts), dim3(256), 0, as_stream(stream), (const float *)xmax, (const float *)xmin, (const float *)dy, (float *)nullptr, G, C, k, relu, partial);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(C), dim3(64), 0, as_stream(stream), partial, (int)parts, C, (double)G * (double)ns, coef, grads);
    k = coef_of(coef, grads, C);
    const unsigned dgrid = (unsigned)((G + per - 1) / per > 16384 ? 16384 : (G + per - 1) / per);
    if (dtype) hipLaunchKernelGGL((bn_pool_dx_kernel<bf16_t>), dim3(dgrid), dim3(256), 0, as_stream(stream), (const bf16_t *)x, (const bf16_t *)dy, (const bf16_t *)xmax, (const bf16_t *)xmin, imax, imin, (bf16_t *)dx, G, ns, C, k, relu);
    else hipLaunchKernelGGL((bn_pool_dx_kernel<float>), dim3(dgrid), dim3(256), 0, as_stream(stream), (const float *)x, (const float *)dy, (const float *)xmax, (const float *)xmin, imax, imin, (float *)dx, G, ns, C, k, relu);
    return check_launch("bn_relu_pool_backward");
}
