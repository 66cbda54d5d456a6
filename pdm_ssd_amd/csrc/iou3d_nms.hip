// Rotated-box BEV overlap / IoU and NMS: the operators of the reference's iou3d_nms
// extension (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu, iou3d_nms.cpp) that detector post-processing calls.
// Boxes are 7 floats [x, y, z, dx, dy, dz, heading].  Same geometry as the reference: rotate the corners, collect
// proper edge intersections and contained corners (1e-2 margin), order them by atan2 about their centroid with the
// same bubble sort, fan area; IoU = overlap / max(sa + sb - overlap, 1e-8); a box is suppressed by an earlier kept
// box iff IoU > thresh (strict).
// MI355X-first difference: the reference copies the N x N/64 suppression mask to the host and reduces it there
// (cudaMemcpy + a CPU loop per call); here a one-wave kernel walks the mask on the device (each lane owns 64-box
// words of the `removed` set), so a call enqueues two kernels and nothing synchronises until the caller reads the count.
#include "common.h"

namespace pdm {

struct P2 { float x, y; };

__device__ __forceinline__ float cross2(P2 a, P2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
__device__ __forceinline__ float fmn(float a, float b) { return a > b ? b : a; }
__device__ __forceinline__ float fmx(float a, float b) { return a > b ? a : b; }

__device__ __forceinline__ bool rects_touch(P2 p1, P2 p2, P2 q1, P2 q2) {
    return fmn(p1.x, p2.x) <= fmx(q1.x, q2.x) && fmn(q1.x, q2.x) <= fmx(p1.x, p2.x) &&
           fmn(p1.y, p2.y) <= fmx(q1.y, q2.y) && fmn(q1.y, q2.y) <= fmx(p1.y, p2.y);
}

__device__ __forceinline__ bool inside_box(const float *box, P2 p) {
    const float margin = 1e-2f;
    const float c = cosf(-box[6]), s = sinf(-box[6]);
    const float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
    const float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
    return fabsf(rx) < box[3] / 2 + margin && fabsf(ry) < box[4] / 2 + margin;
}

__device__ __forceinline__ bool seg_intersection(P2 p1, P2 p0, P2 q1, P2 q0, P2 &ans) {
    if (!rects_touch(p0, p1, q0, q1)) return false;
    const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__device__ __forceinline__ void corners_of(const float *box, P2 *c) {
    const float hx = box[3] / 2, hy = box[4] / 2;
    const float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
    const float ca = cosf(box[6]), sa = sinf(box[6]);
    const float rx[4] = {x1, x2, x2, x1}, ry[4] = {y1, y1, y2, y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dx = rx[k] - box[0], dy = ry[k] - box[1];
        c[k].x = dx * ca + dy * (-sa) + box[0];
        c[k].y = dx * sa + dy * ca + box[1];
    }
    c[4] = c[0];
}

__device__ float box_overlap_bev(const float *a, const float *b) {
    P2 ca[5], cb[5], pts[24];
    corners_of(a, ca);
    corners_of(b, cb);
    int cnt = 0;
    P2 centre{0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 x;
            if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], x)) {
                pts[cnt++] = x;
                centre.x += x.x; centre.y += x.y;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (inside_box(a, cb[k])) { centre.x += cb[k].x; centre.y += cb[k].y; pts[cnt++] = cb[k]; }
        if (inside_box(b, ca[k])) { centre.x += ca[k].x; centre.y += ca[k].y; pts[cnt++] = ca[k]; }
    }
    centre.x /= cnt; centre.y /= cnt;
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (atan2f(pts[i].y - centre.y, pts[i].x - centre.x) > atan2f(pts[i + 1].y - centre.y, pts[i + 1].x - centre.x)) {
                const P2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
            }
    float area = 0;
    for (int k = 0; k < cnt - 1; ++k) {
        const P2 u{pts[k].x - pts[0].x, pts[k].y - pts[0].y}, v{pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y};
        area += cross2(u, v);
    }
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const float *a, const float *b) {
    const float so = box_overlap_bev(a, b);
    return so / fmaxf(a[3] * a[4] + b[3] * b[4] - so, 1e-8f);
}

__device__ __forceinline__ float iou_normal(const float *a, const float *b) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

// out (na, nb): mode 0 overlap area, 1 BEV IoU; 16 x 16 pairs per workgroup, the 16 boxes of each side staged in LDS
__global__ __launch_bounds__(256) void boxes_pairwise_kernel(int mode, int na, const float *__restrict__ boxes_a, int nb,
                                                             const float *__restrict__ boxes_b, float *__restrict__ out) {
    __shared__ float sa[16 * 7], sb[16 * 7];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
    if (threadIdx.x < 112) {
        const int i = threadIdx.x / 7, f = threadIdx.x % 7;
        sa[threadIdx.x] = a0 + i < na ? boxes_a[(size_t)(a0 + i) * 7 + f] : 0.f;
        sb[threadIdx.x] = b0 + i < nb ? boxes_b[(size_t)(b0 + i) * 7 + f] : 0.f;
    }
    __syncthreads();
    const int ai = a0 + ty, bi = b0 + tx;
    if (ai >= na || bi >= nb) return;
    out[(size_t)ai * nb + bi] = mode == 0 ? box_overlap_bev(sa + ty * 7, sb + tx * 7) : iou_bev(sa + ty * 7, sb + tx * 7);
}

__global__ __launch_bounds__(256) void boxes_aligned_overlap_kernel(int n, const float *__restrict__ boxes_a,
                                                                    const float *__restrict__ boxes_b, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[7], b[7];
#pragma unroll
    for (int f = 0; f < 7; ++f) { a[f] = boxes_a[(size_t)i * 7 + f]; b[f] = boxes_b[(size_t)i * 7 + f]; }
    out[i] = box_overlap_bev(a, b);
}

// mask (n, ceil(n/64)) : bit j of word (i, cb) set iff box 64 cb + j (> i) overlaps box i above the threshold
__global__ __launch_bounds__(64) void nms_mask_kernel(int n, float thresh, int normal, const float *__restrict__ boxes,
                                                      unsigned long long *__restrict__ mask) {
    __shared__ float col[64 * 7];
    const int row_start = blockIdx.y, col_start = blockIdx.x;
    const int row_size = min(n - row_start * 64, 64), col_size = min(n - col_start * 64, 64);
    for (int e = threadIdx.x; e < col_size * 7; e += 64) col[e] = boxes[(size_t)col_start * 64 * 7 + e];
    __syncthreads();
    if ((int)threadIdx.x >= row_size) return;
    const int cur = row_start * 64 + threadIdx.x;
    float me[7];
#pragma unroll
    for (int f = 0; f < 7; ++f) me[f] = boxes[(size_t)cur * 7 + f];
    unsigned long long t = 0;
    if (col_start >= row_start) {   // earlier columns never matter: only later boxes can be suppressed by this one
        const int start = row_start == col_start ? threadIdx.x + 1 : 0;
        for (int i = start; i < col_size; ++i) {
            const float v = normal ? iou_normal(me, col + i * 7) : iou_bev(me, col + i * 7);
            if (v > thresh) t |= 1ull << i;
        }
    }
    mask[(size_t)cur * gridDim.x + col_start] = t;
}

// one wave: lane l owns the `removed` words of 64-box blocks l, l + 64, ... ; boxes are visited in score order
__global__ __launch_bounds__(64) void nms_scan_kernel(int n, int col_blocks, const unsigned long long *__restrict__ mask,
                                                      long long *__restrict__ keep, int *__restrict__ num_out) {
    constexpr int WPL = 4;   // words per lane: up to 64 * 4 * 64 = 16384 boxes
    const int lane = threadIdx.x;
    unsigned long long remv[WPL] = {0ull, 0ull, 0ull, 0ull};
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const int nblock = i >> 6, inblock = i & 63;
        const int owner = nblock & 63, slot = nblock >> 6;
        unsigned long long w = slot == 0 ? remv[0] : slot == 1 ? remv[1] : slot == 2 ? remv[2] : remv[3];
        const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)w, owner);
        const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(w >> 32), owner);
        const unsigned long long word = ((unsigned long long)hi << 32) | lo;
        if ((word >> inblock) & 1ull) continue;   // wave-uniform
        if (lane == 0) keep[kept] = i;
        ++kept;
        const unsigned long long *row = mask + (size_t)i * col_blocks;
#pragma unroll
        for (int s_ = 0; s_ < WPL; ++s_) {
            const int cb = s_ * 64 + lane;
            if (cb < col_blocks && cb >= nblock) remv[s_] |= row[cb];
        }
    }
    if (lane == 0) *num_out = kept;
}

// more than 16384 boxes: the same walk with the `removed` words in LDS (col_blocks <= 16384 words = 128 KB)
__global__ __launch_bounds__(64) void nms_scan_lds_kernel(int n, int col_blocks, const unsigned long long *__restrict__ mask,
                                                          long long *__restrict__ keep, int *__restrict__ num_out) {
    extern __shared__ unsigned long long removed[];
    const int lane = threadIdx.x;
    for (int cb = lane; cb < col_blocks; cb += 64) removed[cb] = 0ull;
    __syncthreads();
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const int nblock = i >> 6, inblock = i & 63;
        if ((removed[nblock] >> inblock) & 1ull) continue;   // wave-uniform (same address in every lane)
        if (lane == 0) keep[kept] = i;
        ++kept;
        const unsigned long long *row = mask + (size_t)i * col_blocks;
        for (int cb = nblock + lane; cb < col_blocks; cb += 64) removed[cb] |= row[cb];
        __syncthreads();   // one wave: orders the LDS writes before the next box's read
    }
    if (lane == 0) *num_out = kept;
}

}  // namespace pdm

using namespace pdm;

extern "C" int pdm_boxes_overlap_bev(void *stream, int na, const float *boxes_a, int nb, const float *boxes_b, float *out) {
    PDM_REQUIRE(na >= 0 && nb >= 0, PDM_E_BADARG, "boxes_overlap_bev: negative size");
    if (na == 0 || nb == 0) return 0;
    PDM_REQUIRE(boxes_a && boxes_b && out, PDM_E_BADARG, "boxes_overlap_bev: null pointer");
    hipLaunchKernelGGL(boxes_pairwise_kernel, dim3(divup(nb, 16), divup(na, 16)), dim3(256), 0, as_stream(stream), 0, na, boxes_a,
                       nb, boxes_b, out);
    return check_launch("boxes_overlap_bev");
}

extern "C" int pdm_boxes_iou_bev(void *stream, int na, const float *boxes_a, int nb, const float *boxes_b, float *out) {
    PDM_REQUIRE(na >= 0 && nb >= 0, PDM_E_BADARG, "boxes_iou_bev: negative size");
    if (na == 0 || nb == 0) return 0;
    PDM_REQUIRE(boxes_a && boxes_b && out, PDM_E_BADARG, "boxes_iou_bev: null pointer");
    hipLaunchKernelGGL(boxes_pairwise_kernel, dim3(divup(nb, 16), divup(na, 16)), dim3(256), 0, as_stream(stream), 1, na, boxes_a,
                       nb, boxes_b, out);
    return check_launch("boxes_iou_bev");
}

extern "C" int pdm_boxes_aligned_overlap_bev(void *stream, int n, const float *boxes_a, const float *boxes_b, float *out) {
    PDM_REQUIRE(n >= 0, PDM_E_BADARG, "boxes_aligned_overlap_bev: negative size");
    if (n == 0) return 0;
    PDM_REQUIRE(boxes_a && boxes_b && out, PDM_E_BADARG, "boxes_aligned_overlap_bev: null pointer");
    hipLaunchKernelGGL(boxes_aligned_overlap_kernel, dim3(divup(n, 256)), dim3(256), 0, as_stream(stream), n, boxes_a, boxes_b, out);
    return check_launch("boxes_aligned_overlap_bev");
}

extern "C" size_t pdm_nms_workspace_bytes(int n) {
    if (n <= 0) return 0;
    return (size_t)n * ((n + 63) / 64) * sizeof(unsigned long long);
}

// boxes (n, 7) sorted by descending score; keep (n) int64 receives the kept positions in order, *num_out (device int)
// their count.  normal != 0: axis-aligned footprints (nms_normal_gpu).  Any n the n x n/64 mask workspace allows
// (n <= 1048576; beyond 16384 boxes the suppression walk keeps its bitmap in LDS instead of registers).
extern "C" int pdm_nms(void *stream, int n, const float *boxes, float thresh, int normal, void *workspace,
                       size_t workspace_bytes, long long *keep, int *num_out) {
    PDM_REQUIRE(n >= 0 && n <= 16384 * 64, PDM_E_TOOLARGE, "nms: n=%d (at most 1048576 boxes)", n);
    PDM_REQUIRE(num_out, PDM_E_BADARG, "nms: null pointer");
    if (n == 0) {
        const hipError_t e = hipMemsetAsync(num_out, 0, sizeof(int), as_stream(stream));
        if (e != hipSuccess) { set_error("nms: memset failed"); return (int)e; }
        return 0;
    }
    PDM_REQUIRE(boxes && keep && workspace && workspace_bytes >= pdm_nms_workspace_bytes(n), PDM_E_BADARG,
                "nms: null pointer or workspace of %zu bytes, need %zu", workspace_bytes, pdm_nms_workspace_bytes(n));
    const int cb = (n + 63) / 64;
    unsigned long long *mask = static_cast<unsigned long long *>(workspace);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb), dim3(64), 0, as_stream(stream), n, thresh, normal, boxes, mask);
    int rc = check_launch("nms(mask)");
    if (rc) return rc;
    if (n <= 16384) {
        hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, as_stream(stream), n, cb, mask, keep, num_out);
    } else {
        const size_t lds = (size_t)cb * sizeof(unsigned long long);
        if (lds > 64 * 1024 - 256) {   // granted per device (common.h)
            const int e = grant_lds(reinterpret_cast<const void *>(&nms_scan_lds_kernel), 160 * 1024);
            PDM_REQUIRE(e == 0, PDM_E_TOOLARGE, "nms: cannot obtain %zu bytes of LDS", lds);
        }
        hipLaunchKernelGGL(nms_scan_lds_kernel, dim3(1), dim3(64), lds, as_stream(stream), n, cb, mask, keep, num_out);
    }
    return check_launch("nms(scan)");
}

// ---- points_in_boxes (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:313-336): the first box of the
// sample's list that contains each point, -1 for background.  Boxes staged in LDS in chunks of 128.
namespace pdm {
__global__ __launch_bounds__(256) void points_in_boxes_kernel(int T, int M, const float *__restrict__ boxes,
                                                              const float *__restrict__ pts, int *__restrict__ box_idx) {
    __shared__ float sb[128 * 7];
    const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < M;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) {
        const float *pt = pts + ((size_t)b * M + p) * 3;
        x = pt[0]; y = pt[1]; z = pt[2];
    }
    int found = -1;
    for (int k0 = 0; k0 < T; k0 += 128) {
        const int nk = min(128, T - k0);
        __syncthreads();
        for (int e = threadIdx.x; e < nk * 7; e += blockDim.x) sb[e] = boxes[((size_t)b * T + k0) * 7 + e];
        __syncthreads();
        if (!live || found >= 0) continue;
        for (int k = 0; k < nk; ++k) {
            const float *bx = sb + k * 7;
            if ((double)fabsf(z - bx[2]) > (double)bx[5] / 2.0) continue;
            const float sx = x - bx[0], sy = y - bx[1];
            const float c = cosf(-bx[6]), s = sinf(-bx[6]);
            const float lx = sx * c + sy * (-s), ly = sx * s + sy * c;
            if (fabs((double)lx) < (double)bx[3] / 2.0 + (double)1e-5f && fabs((double)ly) < (double)bx[4] / 2.0 + (double)1e-5f) {
                found = k0 + k;
                break;
            }
        }
    }
    if (live) box_idx[(size_t)b * M + p] = found;
}
}  // namespace pdm

extern "C" int pdm_points_in_boxes(void *stream, int B, int T, int M, const float *boxes, const float *pts, int *box_idx) {
    PDM_REQUIRE(B >= 0 && T >= 0 && M >= 0 && B <= 65535, PDM_E_BADARG, "points_in_boxes: B=%d T=%d M=%d", B, T, M);
    if (B == 0 || M == 0) return 0;
    PDM_REQUIRE(pts && box_idx && (T == 0 || boxes), PDM_E_BADARG, "points_in_boxes: null pointer");
    hipLaunchKernelGGL(pdm::points_in_boxes_kernel, dim3(pdm::divup(M, 256), B), dim3(256), 0, pdm::as_stream(stream), T, M, boxes, pts,
                       box_idx);
    return pdm::check_launch("points_in_boxes");
}
