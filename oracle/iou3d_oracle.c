/*
 * oracle/iou3d_oracle.c — CPU restatement of the reference's rotated-box BEV overlap / IoU / NMS operators
 * (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu, host side iou3d_nms.cpp).  SURVEY.md section 8(f) row N2.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as pointnet2_oracle.c).
 * Parity status: "parity unpinned" — the reference has no tests or vectors for these operators, its kernels are CUDA,
 * and its CPU twin (iou3d_cpu.cpp) includes <cuda.h> / <cuda_runtime_api.h>, which this image lacks, so it cannot
 * be built here either.  Pinned by hand-derived known answers (tests/test_oracle_kat.py): axis-aligned and rotated
 * rectangles whose intersection area is known in closed form.
 *
 * Algorithm followed (iou3d_nms_kernel.cu:104-224): rotate the 4 corners of each box about its centre; collect
 * every proper edge-edge intersection (16 pairs, rejecting by bounding rectangles then by strict opposite-side tests
 * :69-91) and every corner of one box inside the other (tolerance 1e-2, :57-67); order the collected points by
 * atan2 about their centroid with the reference's bubble sort (:200-209); area = |fan sum| / 2.  IoU = overlap /
 * max(sa + sb - overlap, 1e-8) (:227-234).  NMS (:289-339 + iou3d_nms.cpp:137-183): boxes arrive sorted by score;
 * box i is kept iff no earlier KEPT box j has iou(j, i) > thresh (strict).
 * fp32 throughout; cosf / sinf / atan2f of the C library (the HIP kernels use the device library's: last-bit
 * differences are possible, tests compare areas to 1e-5 relative and decisions away from the threshold).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } P2;

static inline float cross2(P2 a, P2 b) { return a.x * b.y - a.y * b.x; }
static inline float cross3(P2 p1, P2 p2, P2 p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
static inline float fmn(float a, float b) { return a > b ? b : a; }
static inline float fmx(float a, float b) { return a > b ? a : b; }

static int rects_touch(P2 p1, P2 p2, P2 q1, P2 q2) {
    return fmn(p1.x, p2.x) <= fmx(q1.x, q2.x) && fmn(q1.x, q2.x) <= fmx(p1.x, p2.x) &&
           fmn(p1.y, p2.y) <= fmx(q1.y, q2.y) && fmn(q1.y, q2.y) <= fmx(p1.y, p2.y);
}

/* corner p inside box (x, y, z, dx, dy, dz, heading) with the reference's 1e-2 margin (:57-67) */
static int inside_box(const float *box, P2 p) {
    const float margin = 1e-2f;
    const float c = cosf(-box[6]), s = sinf(-box[6]);
    const float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
    const float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
    return fabsf(rx) < box[3] / 2 + margin && fabsf(ry) < box[4] / 2 + margin;
}

/* proper intersection of segments p0p1 and q0q1 (:69-100) */
static int seg_intersection(P2 p1, P2 p0, P2 q1, P2 q0, P2 *ans) {
    if (!rects_touch(p0, p1, q0, q1)) return 0;
    const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

static void corners_of(const float *box, P2 *c) {
    const float hx = box[3] / 2, hy = box[4] / 2;
    const float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
    const float ca = cosf(box[6]), sa = sinf(box[6]);
    const P2 raw[4] = {{x1, y1}, {x2, y1}, {x2, y2}, {x1, y2}};
    for (int k = 0; k < 4; ++k) {
        const float dx = raw[k].x - box[0], dy = raw[k].y - box[1];
        c[k].x = dx * ca + dy * (-sa) + box[0];
        c[k].y = dx * sa + dy * ca + box[1];
    }
    c[4] = c[0];
}

float oracle_box_overlap_bev(const float *a, const float *b) {
    P2 ca[5], cb[5], pts[24];
    corners_of(a, ca);
    corners_of(b, cb);
    int cnt = 0;
    P2 centre = {0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &pts[cnt])) {
                centre.x += pts[cnt].x; centre.y += pts[cnt].y;
                ++cnt;
            }
    for (int k = 0; k < 4; ++k) {
        if (inside_box(a, cb[k])) { centre.x += cb[k].x; centre.y += cb[k].y; pts[cnt++] = cb[k]; }
        if (inside_box(b, ca[k])) { centre.x += ca[k].x; centre.y += ca[k].y; pts[cnt++] = ca[k]; }
    }
    centre.x /= cnt; centre.y /= cnt;
    for (int j = 0; j < cnt - 1; ++j)      /* the reference's bubble sort: descending... swaps when angle(i) > angle(i+1) */
        for (int i = 0; i < cnt - j - 1; ++i)
            if (atan2f(pts[i].y - centre.y, pts[i].x - centre.x) > atan2f(pts[i + 1].y - centre.y, pts[i + 1].x - centre.x)) {
                const P2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
            }
    float area = 0;
    for (int k = 0; k < cnt - 1; ++k) {
        const P2 u = {pts[k].x - pts[0].x, pts[k].y - pts[0].y}, v = {pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y};
        area += cross2(u, v);
    }
    return fabsf(area) / 2.0f;
}

float oracle_iou_bev(const float *a, const float *b) {
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    const float so = oracle_box_overlap_bev(a, b);
    return so / fmaxf(sa + sb - so, 1e-8f);
}

/* iou3d_nms_kernel.cu:342-353: axis-aligned footprints (heading ignored) */
float oracle_iou_normal(const float *a, const float *b) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

/* mode 0: overlap area, 1: BEV IoU */
int oracle_boxes_pairwise_bev(int mode, int na, const float *boxes_a, int nb, const float *boxes_b, float *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j)
            out[(size_t)i * nb + j] = mode == 0 ? oracle_box_overlap_bev(boxes_a + i * 7, boxes_b + j * 7)
                                                : oracle_iou_bev(boxes_a + i * 7, boxes_b + j * 7);
    return 0;
}

int oracle_boxes_aligned_overlap_bev(int n, const float *boxes_a, const float *boxes_b, float *out) {
    for (int i = 0; i < n; ++i) out[i] = oracle_box_overlap_bev(boxes_a + i * 7, boxes_b + i * 7);
    return 0;
}

/* greedy NMS over boxes already sorted by descending score; keep receives indices into that order.
 * normal = 0: rotated BEV IoU, 1: axis-aligned IoU.  Returns the number kept. */
int oracle_nms(int normal, int n, const float *boxes, float thresh, long long *keep) {
    char *removed = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        if (removed[i]) continue;
        keep[kept++] = i;
        for (int j = i + 1; j < n; ++j) {
            if (removed[j]) continue;
            const float v = normal ? oracle_iou_normal(boxes + i * 7, boxes + j * 7) : oracle_iou_bev(boxes + i * 7, boxes + j * 7);
            if (v > thresh) removed[j] = 1;
        }
    }
    free(removed);
    return kept;
}

/* pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-36, 313-336 — for every point the FIRST box (in list
 * order) that contains it, -1 for background.  Inside: |z - cz| <= dz/2 and, after rotating the offset by -heading,
 * |lx| < dx/2 + 1e-5 and |ly| < dy/2 + 1e-5 (the reference evaluates these comparisons in double: dz / 2.0, the
 * float MARGIN promoted).  boxes (B, T, 7), pts (B, M, 3) -> box_idx (B, M). */
int oracle_points_in_boxes(int B, int T, int M, const float *boxes, const float *pts, int *box_idx) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < M; ++p) {
            const float *pt = pts + ((size_t)b * M + p) * 3;
            int found = -1;
            for (int k = 0; k < T && found < 0; ++k) {
                const float *bx = boxes + ((size_t)b * T + k) * 7;
                if ((double)fabsf(pt[2] - bx[2]) > (double)bx[5] / 2.0) continue;
                const float sx = pt[0] - bx[0], sy = pt[1] - bx[1];
                const float c = cosf(-bx[6]), s = sinf(-bx[6]);
                const float lx = sx * c + sy * (-s), ly = sx * s + sy * c;
                const float margin = 1e-5f;
                if (fabs((double)lx) < (double)bx[3] / 2.0 + (double)margin && fabs((double)ly) < (double)bx[4] / 2.0 + (double)margin) found = k;
            }
            box_idx[(size_t)b * M + p] = found;
        }
    return 0;
}
