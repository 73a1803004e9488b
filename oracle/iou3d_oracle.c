/*
 * oracle/iou3d_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the rotated-box BEV overlap / IoU / 3-D NMS operators of
 * `pcdet.ops.iou3d_nms.iou3d_nms_cuda` (reference:
 * OpenPCDet/pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu, iou3d_cpu.cpp, iou3d_nms.cpp).
 *
 * Parity status: box overlap / iou_bev are PINNED bit-for-bit against the compiled
 * reference iou3d_cpu.cpp (oracle/_ref, see oracle/Makefile and
 * tests/test_oracle_vs_ref.py) and against the committed golden vectors generated
 * from it (tests/golden/iou_bev_cpu_ref.npz).  The NMS kernels have no reference
 * CPU form; they are restated from iou3d_nms_kernel.cu:237-247,280-385 and the
 * host scan iou3d_nms.cpp:121-134 on top of the pinned box overlap.
 *
 * The reference IoU is an APPROXIMATION (corner-in-box test with a 1 cm margin);
 * parity is to that approximation, step for step: same candidate order, same
 * margin, same bubble sort, same shoelace fan from the first sorted vertex.
 *
 * All arithmetic is fp32 in source order (build with -ffp-contract=off).
 * Boxes are (x, y, z, dx, dy, dz, heading), 7 floats.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BOX_STRIDE 7
#define NMS_TILE 64 /* sizeof(unsigned long long)*8, iou3d_nms_kernel.cu:13 */

typedef struct { float x, y; } pt2;

/* iou3d_nms_kernel.cu:40-42 */
static inline float cross3(pt2 p1, pt2 p2, pt2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

/* iou3d_nms_kernel.cu:36-38 */
static inline float cross2(pt2 a, pt2 b) { return a.x * b.y - a.y * b.x; }

static inline float fmin2(float a, float b) { return a > b ? b : a; }
static inline float fmax2(float a, float b) { return a > b ? a : b; }

/* Bounding-box rejection of two segments, iou3d_nms_kernel.cu:44-50 (<= on all four). */
static inline int seg_bbox_touch(pt2 p1, pt2 p2, pt2 q1, pt2 q2) {
  return fmin2(p1.x, p2.x) <= fmax2(q1.x, q2.x) &&
         fmin2(q1.x, q2.x) <= fmax2(p1.x, p2.x) &&
         fmin2(p1.y, p2.y) <= fmax2(q1.y, q2.y) &&
         fmin2(q1.y, q2.y) <= fmax2(p1.y, p2.y);
}

/* Corner-in-box with the reference's 1e-2 margin, iou3d_nms_kernel.cu:52-62.
 * cos/sin of -heading are recomputed per call, as in the reference. */
static inline int corner_in_box(const float *box, pt2 p) {
  const float MARGIN = 1e-2f;
  const float cx = box[0], cy = box[1];
  const float ac = cosf(-box[6]), as = sinf(-box[6]);
  const float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
  const float ry = (p.x - cx) * as + (p.y - cy) * ac;
  return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}

/* Segment/segment crossing point, iou3d_nms_kernel.cu:64-93.  (p1,p0) is an edge of
 * box a, (q1,q0) an edge of box b. */
static inline int seg_cross(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2 *ans) {
  if (!seg_bbox_touch(p0, p1, q0, q1)) return 0;
  const float s1 = cross3(q0, p1, p0);
  const float s2 = cross3(p1, q1, p0);
  const float s3 = cross3(p0, q1, q0);
  const float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  const float s5 = cross3(q1, p1, p0);
  /* EPS is a double literal in the device file and a float const in the CPU file;
   * no float lies strictly between (float)1e-8 and 1e-8, so both agree. */
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

/* iou3d_nms_kernel.cu:95-99 */
static inline pt2 rot_about(pt2 c, float ac, float as, pt2 p) {
  pt2 r;
  r.x = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
  r.y = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
  return r;
}

/* BEV intersection area of two rotated rectangles.
 * Follows iou3d_nms_kernel.cu:105-226 (== iou3d_cpu.cpp:128-220). */
float iou3do_box_overlap(const float *box_a, const float *box_b) {
  const float a_angle = box_a[6], b_angle = box_b[6];
  const float a_dx_half = box_a[3] / 2, b_dx_half = box_b[3] / 2;
  const float a_dy_half = box_a[4] / 2, b_dy_half = box_b[4] / 2;
  const float a_x1 = box_a[0] - a_dx_half, a_y1 = box_a[1] - a_dy_half;
  const float a_x2 = box_a[0] + a_dx_half, a_y2 = box_a[1] + a_dy_half;
  const float b_x1 = box_b[0] - b_dx_half, b_y1 = box_b[1] - b_dy_half;
  const float b_x2 = box_b[0] + b_dx_half, b_y2 = box_b[1] + b_dy_half;

  const pt2 ca = {box_a[0], box_a[1]}, cb = {box_b[0], box_b[1]};
  pt2 A[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  pt2 Bc[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};

  const float a_cos = cosf(a_angle), a_sin = sinf(a_angle);
  const float b_cos = cosf(b_angle), b_sin = sinf(b_angle);
  for (int k = 0; k < 4; ++k) {
    A[k] = rot_about(ca, a_cos, a_sin, A[k]);
    Bc[k] = rot_about(cb, b_cos, b_sin, Bc[k]);
  }
  A[4] = A[0];
  Bc[4] = Bc[0];

  pt2 poly[16];
  pt2 ctr = {0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (seg_cross(A[i + 1], A[i], Bc[j + 1], Bc[j], &poly[cnt])) {
        ctr.x = ctr.x + poly[cnt].x;
        ctr.y = ctr.y + poly[cnt].y;
        ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (corner_in_box(box_a, Bc[k])) {
      ctr.x = ctr.x + Bc[k].x;
      ctr.y = ctr.y + Bc[k].y;
      poly[cnt++] = Bc[k];
    }
    if (corner_in_box(box_b, A[k])) {
      ctr.x = ctr.x + A[k].x;
      ctr.y = ctr.y + A[k].y;
      poly[cnt++] = A[k];
    }
  }
  ctr.x /= cnt; /* 0/0 when cnt == 0; unused then */
  ctr.y /= cnt;

  /* bubble sort ascending by atan2 about the centroid, '>' comparison (:201-210) */
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (atan2f(poly[i].y - ctr.y, poly[i].x - ctr.x) >
          atan2f(poly[i + 1].y - ctr.y, poly[i + 1].x - ctr.x)) {
        const pt2 t = poly[i];
        poly[i] = poly[i + 1];
        poly[i + 1] = t;
      }

  float area = 0;
  for (int k = 0; k < cnt - 1; ++k) {
    const pt2 u = {poly[k].x - poly[0].x, poly[k].y - poly[0].y};
    const pt2 v = {poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y};
    area += cross2(u, v);
  }
  return (float)(fabsf(area) / 2.0);
}

/* iou3d_nms_kernel.cu:228-235 */
float iou3do_iou_bev(const float *box_a, const float *box_b) {
  const float sa = box_a[3] * box_a[4];
  const float sb = box_b[3] * box_b[4];
  const float s_overlap = iou3do_box_overlap(box_a, box_b);
  return s_overlap / fmaxf(sa + sb - s_overlap, 1e-8f);
}

/* iou3d_nms_kernel.cu:237-247 (fork-local: NMS uses this 3-D IoU, not BEV IoU) */
float iou3do_iou_bev_3d(const float *box_a, const float *box_b) {
  const float sa = box_a[3] * box_a[4] * box_a[5];
  const float sb = box_b[3] * box_b[4] * box_b[5];
  const float top = fmaxf(box_a[2] - box_a[5] / 2, box_b[2] - box_b[5] / 2);
  const float bottom = fminf(box_a[2] + box_a[5] / 2, box_b[2] + box_b[5] / 2);
  const float height = fmaxf(bottom - top, 0.f);
  const float s_overlap = iou3do_box_overlap(box_a, box_b) * height;
  return s_overlap / fmaxf(sa + sb - s_overlap, 1e-8f);
}

/* iou3d_nms_kernel.cu:327-338 (axis-aligned BEV IoU, ignores heading and z) */
float iou3do_iou_normal(const float *a, const float *b) {
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2);
  const float right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2);
  const float bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  const float interS = width * height;
  const float Sa = a[3] * a[4];
  const float Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, 1e-8f);
}

/* (N,M) BEV overlap matrix: boxes_overlap_kernel, iou3d_nms_kernel.cu:249-262 */
void iou3do_boxes_overlap_bev(int na, const float *boxes_a, int nb,
                              const float *boxes_b, float *ans) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      ans[(size_t)i * nb + j] =
          iou3do_box_overlap(boxes_a + i * BOX_STRIDE, boxes_b + j * BOX_STRIDE);
}

/* (N,M) BEV IoU matrix: boxes_iou_bev_kernel :264-278 == boxes_iou_bev_cpu
 * iou3d_cpu.cpp:232-252 */
void iou3do_boxes_iou_bev(int na, const float *boxes_a, int nb,
                          const float *boxes_b, float *ans) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      ans[(size_t)i * nb + j] =
          iou3do_iou_bev(boxes_a + i * BOX_STRIDE, boxes_b + j * BOX_STRIDE);
}

/* 3-D IoU matrix as the Python wrapper boxes_iou3d_gpu composes it around the
 * overlap kernel (iou3d_nms_utils.py:48-81): z-overlap, volumes, and a 1e-6 clamp
 * on the denominator (NOT the kernel's 1e-8). */
void iou3do_boxes_iou3d(int na, const float *boxes_a, int nb,
                        const float *boxes_b, float *ans) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < na; ++i) {
    const float *a = boxes_a + i * BOX_STRIDE;
    const float a_max = a[2] + a[5] / 2, a_min = a[2] - a[5] / 2;
    const float vol_a = a[3] * a[4] * a[5];
    for (int j = 0; j < nb; ++j) {
      const float *bb = boxes_b + j * BOX_STRIDE;
      const float b_max = bb[2] + bb[5] / 2, b_min = bb[2] - bb[5] / 2;
      const float vol_b = bb[3] * bb[4] * bb[5];
      const float ov_bev = iou3do_box_overlap(a, bb);
      const float max_of_min = a_min > b_min ? a_min : b_min;
      const float min_of_max = a_max < b_max ? a_max : b_max;
      float h = min_of_max - max_of_min;
      if (h < 0.f) h = 0.f;
      const float ov3d = ov_bev * h;
      float den = vol_a + vol_b - ov3d;
      if (den < 1e-6f) den = 1e-6f;
      ans[(size_t)i * nb + j] = ov3d / den;
    }
  }
}

/* Suppression bitmask, nms_kernel :280-324 / nms_normal_kernel :341-385.
 * mask is (N, ceil(N/64)) u64; bit i of word [r, cb] = IoU(r, cb*64+i) > thresh;
 * on diagonal tiles only columns strictly after the row are tested. */
static void nms_mask(const float *boxes, int n, float thresh, int normal,
                     unsigned long long *mask) {
  const int col_blocks = n / NMS_TILE + (n % NMS_TILE > 0);
#pragma omp parallel for schedule(dynamic, 8)
  for (int r = 0; r < n; ++r) {
    const int rb = r / NMS_TILE, rl = r % NMS_TILE;
    for (int cb = 0; cb < col_blocks; ++cb) {
      int col_size = n - cb * NMS_TILE;
      if (col_size > NMS_TILE) col_size = NMS_TILE;
      unsigned long long t = 0;
      for (int i = (rb == cb) ? rl + 1 : 0; i < col_size; ++i) {
        const float *cbx = boxes + (size_t)(cb * NMS_TILE + i) * BOX_STRIDE;
        const float v = normal ? iou3do_iou_normal(boxes + (size_t)r * BOX_STRIDE, cbx)
                               : iou3do_iou_bev_3d(boxes + (size_t)r * BOX_STRIDE, cbx);
        if (v > thresh) t |= 1ULL << i;
      }
      mask[(size_t)r * col_blocks + cb] = t;
    }
  }
}

/* Greedy scan over the mask, iou3d_nms.cpp:121-134.  Returns the kept count. */
static int nms_scan(const unsigned long long *mask, int n, long long *keep) {
  const int col_blocks = n / NMS_TILE + (n % NMS_TILE > 0);
  unsigned long long *remv =
      (unsigned long long *)calloc((size_t)(col_blocks > 0 ? col_blocks : 1),
                                   sizeof(unsigned long long));
  int num_to_keep = 0;
  for (int i = 0; i < n; ++i) {
    const int nblock = i / NMS_TILE, inblock = i % NMS_TILE;
    if (!(remv[nblock] & (1ULL << inblock))) {
      keep[num_to_keep++] = i;
      const unsigned long long *p = mask + (size_t)i * col_blocks;
      for (int j = nblock; j < col_blocks; ++j) remv[j] |= p[j];
    }
  }
  free(remv);
  return num_to_keep;
}

/* boxes must already be sorted by score, descending (iou3d_nms_utils.py:92-96).
 * keep receives int64 indices (the upstream-OpenPCDet contract; the fork's C++
 * reads the LongTensor as int32, iou3d_nms.cpp:98 -- see SURVEY section 0 defect 2). */
int iou3do_nms(const float *boxes, int n, float thresh, long long *keep,
               unsigned long long *mask_out) {
  const int col_blocks = n / NMS_TILE + (n % NMS_TILE > 0);
  unsigned long long *mask = mask_out
      ? mask_out
      : (unsigned long long *)malloc(sizeof(unsigned long long) *
                                     (size_t)(n > 0 ? n : 1) * (col_blocks > 0 ? col_blocks : 1));
  nms_mask(boxes, n, thresh, 0, mask);
  const int k = nms_scan(mask, n, keep);
  if (!mask_out) free(mask);
  return k;
}

int iou3do_nms_normal(const float *boxes, int n, float thresh, long long *keep,
                      unsigned long long *mask_out) {
  const int col_blocks = n / NMS_TILE + (n % NMS_TILE > 0);
  unsigned long long *mask = mask_out
      ? mask_out
      : (unsigned long long *)malloc(sizeof(unsigned long long) *
                                     (size_t)(n > 0 ? n : 1) * (col_blocks > 0 ? col_blocks : 1));
  nms_mask(boxes, n, thresh, 1, mask);
  const int k = nms_scan(mask, n, keep);
  if (!mask_out) free(mask);
  return k;
}
