// 3dioumatch_amd/csrc/box_geom.h -- rotated-rectangle intersection, shared by the gfx950
// kernels (iou3d.hip) and the host implementation of boxes_iou_bev_cpu.
//
// Semantics: reference iou3d_nms_kernel.cu:36-247 (== iou3d_cpu.cpp:59-229), SURVEY App. A.8.
// The reference result is an approximation (corner test with a 1 cm margin, bubble sort by
// atan2, shoelace fan from the first sorted vertex); parity means reproducing exactly that
// sequence of fp32 operations.  What is restructured here WITHOUT changing any value:
//   * everything that depends on one box only (rotated corners, cos/sin(-heading), the
//     margin-inflated half extents) is computed once per box (BoxPre), not once per pair
//     or -- as the reference does for cos/sin -- once per corner test;
//   * the polar angle of each polygon vertex is computed once (the reference recomputes two
//     atan2 per comparison of its bubble sort); the sort itself is the same bubble sort with
//     the same '>' comparison, so the permutation is identical;
//   * a conservative bounding-circle test returns the exact result 0 for far-apart pairs
//     before any of the above (such pairs have no crossing and no corner inside the margin,
//     for which the reference computes area 0);
//   * the <=16 candidate vertices live in a caller-provided store (LDS on the device,
//     strided per lane; a stack array on the host) because they are indexed dynamically.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define BG_HD __host__ __device__ __forceinline__
#else
#define BG_HD inline
#endif

#pragma clang fp contract(off)

namespace boxgeom {

constexpr int kMaxPoly = 16;  // cross_points[16], iou3d_nms_kernel.cu:156

struct BoxPre {
  float cx, cy;          // centre
  float px[4], py[4];    // corners rotated by +heading about the centre
  float ncos, nsin;      // cos(-heading), sin(-heading)  (check_in_box2d, :57)
  float hxm, hym;        // dx/2 + 1e-2, dy/2 + 1e-2      (check_in_box2d, :61)
  float rad;             // half diagonal (for the conservative reject only)
};

BG_HD void box_prepare(const float *box, BoxPre &o) {
  const float ang = box[6];
  const float dxh = box[3] / 2, dyh = box[4] / 2;
  const float x1 = box[0] - dxh, y1 = box[1] - dyh;
  const float x2 = box[0] + dxh, y2 = box[1] + dyh;
  o.cx = box[0];
  o.cy = box[1];
  const float ac = cosf(ang), as = sinf(ang);
  const float qx[4] = {x1, x2, x2, x1};
  const float qy[4] = {y1, y1, y2, y2};
  for (int k = 0; k < 4; ++k) {  // rotate_around_center, :95-99
    o.px[k] = (qx[k] - o.cx) * ac + (qy[k] - o.cy) * (-as) + o.cx;
    o.py[k] = (qx[k] - o.cx) * as + (qy[k] - o.cy) * ac + o.cy;
  }
  o.ncos = cosf(-ang);
  o.nsin = sinf(-ang);
  o.hxm = box[3] / 2 + 1e-2f;
  o.hym = box[4] / 2 + 1e-2f;
  o.rad = sqrtf(dxh * dxh + dyh * dyh);
}

BG_HD float cross3(float p1x, float p1y, float p2x, float p2y, float p0x, float p0y) {
  return (p1x - p0x) * (p2y - p0y) - (p2x - p0x) * (p1y - p0y);  // :40-42
}

BG_HD float fmin2(float a, float b) { return a > b ? b : a; }
BG_HD float fmax2(float a, float b) { return a > b ? a : b; }

// intersection(), :64-93.  (p1,p0): edge of a; (q1,q0): edge of b.
BG_HD bool seg_cross(float p1x, float p1y, float p0x, float p0y, float q1x, float q1y,
                     float q0x, float q0y, float &ax, float &ay) {
  const bool touch = fmin2(p0x, p1x) <= fmax2(q0x, q1x) && fmin2(q0x, q1x) <= fmax2(p0x, p1x) &&
                     fmin2(p0y, p1y) <= fmax2(q0y, q1y) && fmin2(q0y, q1y) <= fmax2(p0y, p1y);
  if (!touch) return false;
  const float s1 = cross3(q0x, q0y, p1x, p1y, p0x, p0y);
  const float s2 = cross3(p1x, p1y, q1x, q1y, p0x, p0y);
  const float s3 = cross3(p0x, p0y, q1x, q1y, q0x, q0y);
  const float s4 = cross3(q1x, q1y, p1x, p1y, q0x, q0y);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = cross3(q1x, q1y, p1x, p1y, p0x, p0y);
  if (fabsf(s5 - s1) > 1e-8f) {
    ax = (s5 * q0x - s1 * q1x) / (s5 - s1);
    ay = (s5 * q0y - s1 * q1y) / (s5 - s1);
  } else {
    const float a0 = p0y - p1y, b0 = p1x - p0x, c0 = p0x * p1y - p1x * p0y;
    const float a1 = q0y - q1y, b1 = q1x - q0x, c1 = q0x * q1y - q1x * q0y;
    const float D = a0 * b1 - a1 * b0;
    ax = (b0 * c1 - b1 * c0) / D;
    ay = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

// check_in_box2d(), :52-62, with the per-box terms taken from BoxPre
BG_HD bool corner_in_box(const BoxPre &bx, float px, float py) {
  const float rx = (px - bx.cx) * bx.ncos + (py - bx.cy) * (-bx.nsin);
  const float ry = (px - bx.cx) * bx.nsin + (py - bx.cy) * bx.ncos;
  return fabsf(rx) < bx.hxm && fabsf(ry) < bx.hym;
}

// true when the pair certainly has an empty candidate set (=> the reference returns 0)
BG_HD bool far_apart(const BoxPre &a, const BoxPre &b) {
  const float dx = a.cx - b.cx, dy = a.cy - b.cy;
  const float d = sqrtf(dx * dx + dy * dy);
  return d > (a.rad + b.rad) * 1.001f + 0.05f;  // false for NaN => full path
}

// box_overlap(), :105-226.  Store provides x(i), y(i), a(i) lvalues for i < kMaxPoly.
template <class Store>
BG_HD float overlap_area(const BoxPre &A, const BoxPre &B, Store &st) {
  if (far_apart(A, B)) return 0.f;
  int cnt = 0;
  float sx = 0.f, sy = 0.f;
  for (int i = 0; i < 4; ++i) {
    const int i1 = (i + 1) & 3;
    for (int j = 0; j < 4; ++j) {
      const int j1 = (j + 1) & 3;
      float ax, ay;
      if (seg_cross(A.px[i1], A.py[i1], A.px[i], A.py[i], B.px[j1], B.py[j1], B.px[j], B.py[j],
                    ax, ay)) {
        sx = sx + ax;
        sy = sy + ay;
        if (cnt < kMaxPoly) { st.x(cnt) = ax; st.y(cnt) = ay; }
        ++cnt;
      }
    }
  }
  for (int k = 0; k < 4; ++k) {
    if (corner_in_box(A, B.px[k], B.py[k])) {
      sx = sx + B.px[k];
      sy = sy + B.py[k];
      if (cnt < kMaxPoly) { st.x(cnt) = B.px[k]; st.y(cnt) = B.py[k]; }
      ++cnt;
    }
    if (corner_in_box(B, A.px[k], A.py[k])) {
      sx = sx + A.px[k];
      sy = sy + A.py[k];
      if (cnt < kMaxPoly) { st.x(cnt) = A.px[k]; st.y(cnt) = A.py[k]; }
      ++cnt;
    }
  }
  if (cnt == 0) return 0.f;
  const float mx = sx / cnt, my = sy / cnt;
  if (cnt > kMaxPoly) cnt = kMaxPoly;  // the reference would overrun its array here
  for (int i = 0; i < cnt; ++i) st.a(i) = atan2f(st.y(i) - my, st.x(i) - mx);
  for (int j = 0; j < cnt - 1; ++j)        // bubble sort, '>' (:201-210)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (st.a(i) > st.a(i + 1)) {
        const float tx = st.x(i), ty = st.y(i), ta = st.a(i);
        st.x(i) = st.x(i + 1); st.y(i) = st.y(i + 1); st.a(i) = st.a(i + 1);
        st.x(i + 1) = tx; st.y(i + 1) = ty; st.a(i + 1) = ta;
      }
  float area = 0.f;
  const float x0 = st.x(0), y0 = st.y(0);
  for (int k = 0; k < cnt - 1; ++k) {
    const float ux = st.x(k) - x0, uy = st.y(k) - y0;
    const float vx = st.x(k + 1) - x0, vy = st.y(k + 1) - y0;
    area += ux * vy - uy * vx;  // cross(a, b), :36-38
  }
  return fabsf(area) / 2.0f;
}

}  // namespace boxgeom
