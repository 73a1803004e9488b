// 3dioumatch_amd/csrc/eval_obb_iou.hip -- oriented-box IoU of the AP evaluation on the device
// (gfx950), SURVEY section 8(f) rank 4.
//
// What it replaces: the innermost loops of eval_det_cls (utils/eval_det.py:128-141).  For every
// detection, in score order, the reference calls get_iou_obb = box3d_iou
// (utils/box_util.py:112-137) against each ground-truth box of the same scan and class -- pure
// Python: Sutherland-Hodgman clipping of the two footprints (polygon_clip, box_util.py:23-69), a
// scipy ConvexHull for the area of the clipped polygon (:77-88), the overlap of the vertical
// extents and box3d_vol (:91-96) -- and keeps (ovmax, jmax) under a strict `>` update.  The
// detections are independent of each other until the greedy TP/FP marking, so the IoUs and the
// per-detection maximum are computed here in one launch and only the marking stays on the host.
//
// Arithmetic: float32 corners promoted to float64 (the reference's `.astype(float)`,
// eval_det.py:130-132), every expression evaluated in the reference's order (-ffp-contract=off),
// so the strict inside() tests of the clipping take the same branches.  The area of the clipped
// polygon is the shoelace sum (the polygon is convex, so this is the hull area the reference gets
// from qhull, to rounding); a clipped polygon with fewer than 3 vertices has area 0 (the
// reference raises QhullError there).
//
// One thread per detection (best_match) or per pair (matrix); the polygons live in registers /
// scratch (<= 8 vertices: a quad gains at most one vertex per clipping edge).
#include "common.h"

namespace {

struct P2 { double x, y; };

// polygon_clip (box_util.py:23-69) of quad `subj` by convex quad `clip`, then its area.
__device__ double clipped_area(const P2 *subj, const P2 *clip) {
  P2 buf[2][10];  // ping-pong: the output list of one clipping edge is the input of the next
  int cur = 0, n_out = 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) buf[0][i] = subj[i];
  P2 cp1 = clip[3];
  for (int c = 0; c < 4; ++c) {
    const P2 cp2 = clip[c];
    const P2 *in = buf[cur];
    P2 *out = buf[cur ^ 1];
    const int n_in = n_out;
    n_out = 0;
    P2 s = in[n_in - 1];
    const double ex = cp2.x - cp1.x, ey = cp2.y - cp1.y;
    bool s_in = ex * (s.y - cp1.y) > ey * (s.x - cp1.x);
    for (int i = 0; i < n_in; ++i) {
      const P2 e = in[i];
      const bool e_in = ex * (e.y - cp1.y) > ey * (e.x - cp1.x);
      if (e_in != s_in) {  // computeIntersection (box_util.py:40-46)
        const double dcx = cp1.x - cp2.x, dcy = cp1.y - cp2.y;
        const double dpx = s.x - e.x, dpy = s.y - e.y;
        const double n1 = cp1.x * cp2.y - cp1.y * cp2.x;
        const double n2 = s.x * e.y - s.y * e.x;
        const double n3 = 1.0 / (dcx * dpy - dcy * dpx);
        P2 q;
        q.x = (n1 * dpx - n2 * dcx) * n3;
        q.y = (n1 * dpy - n2 * dcy) * n3;
        if (n_out < 10) out[n_out++] = q;
      }
      if (e_in && n_out < 10) out[n_out++] = e;
      s = e;
      s_in = e_in;
    }
    cp1 = cp2;
    cur ^= 1;
    if (n_out == 0) return 0.0;
  }
  const P2 *out = buf[cur];
  if (n_out < 3) return 0.0;
  double acc = 0.0;
  for (int i = 0; i < n_out; ++i) {
    const P2 a = out[i], b = out[i + 1 == n_out ? 0 : i + 1];
    acc += a.x * b.y - a.y * b.x;
  }
  return 0.5 * fabs(acc);
}

__device__ __forceinline__ double edge_len(const float *c, int i, int j) {
  const double dx = (double)c[i * 3 + 0] - (double)c[j * 3 + 0];
  const double dy = (double)c[i * 3 + 1] - (double)c[j * 3 + 1];
  const double dz = (double)c[i * 3 + 2] - (double)c[j * 3 + 2];
  return sqrt((dx * dx + dy * dy) + dz * dz);
}

// box3d_iou (box_util.py:112-137), the 3-D value; c1, c2: (8,3) float32 corners, up = -Y.
__device__ double box3d_iou(const float *c1, const float *c2) {
  P2 r1[4], r2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // footprints, vertices 3,2,1,0 (counter-clockwise)
    r1[k].x = c1[(3 - k) * 3 + 0]; r1[k].y = c1[(3 - k) * 3 + 2];
    r2[k].x = c2[(3 - k) * 3 + 0]; r2[k].y = c2[(3 - k) * 3 + 2];
  }
  const double inter_area = clipped_area(r1, r2);
  const double ymax = fmin((double)c1[1], (double)c2[1]);
  const double ymin = fmax((double)c1[4 * 3 + 1], (double)c2[4 * 3 + 1]);
  const double inter_vol = inter_area * fmax(0.0, ymax - ymin);
  const double vol1 = edge_len(c1, 0, 1) * edge_len(c1, 1, 2) * edge_len(c1, 0, 4);
  const double vol2 = edge_len(c2, 0, 1) * edge_len(c2, 1, 2) * edge_len(c2, 0, 4);
  return inter_vol / (vol1 + vol2 - inter_vol);
}

__device__ __forceinline__ void load_box(const float *src, float *dst) {
  const float4 *s = reinterpret_cast<const float4 *>(src);  // 24 floats, 96-byte stride
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const float4 v = s[q];
    dst[q * 4 + 0] = v.x; dst[q * 4 + 1] = v.y; dst[q * 4 + 2] = v.z; dst[q * 4 + 3] = v.w;
  }
}

__global__ __launch_bounds__(128)
void corners_iou_matrix_kernel(int n, int m, const float *__restrict__ a,
                               const float *__restrict__ b, double *__restrict__ iou) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long long)n * m) return;
  float ca[24], cb[24];
  load_box(a + (p / m) * 24, ca);
  load_box(b + (p % m) * 24, cb);
  iou[p] = box3d_iou(ca, cb);
}

// eval_det.py:128-141: ovmax = -inf; for j: if iou > ovmax: ovmax = iou, jmax = j
__global__ __launch_bounds__(128)
void corners_best_match_kernel(int nd, const float *__restrict__ det, const int *__restrict__ gt_begin,
                               const int *__restrict__ gt_count, const float *__restrict__ gt,
                               double *__restrict__ ovmax, int *__restrict__ jmax) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= nd) return;
  float cd[24], cg[24];
  load_box(det + (long long)d * 24, cd);
  const int g0 = gt_begin[d], cnt = gt_count[d];
  double best = -INFINITY;
  int bj = -1;
  for (int j = 0; j < cnt; ++j) {
    load_box(gt + (long long)(g0 + j) * 24, cg);
    const double v = box3d_iou(cd, cg);
    if (v > best) { best = v; bj = j; }
  }
  ovmax[d] = best;
  jmax[d] = bj;
}

}  // namespace

// iou (n,m) f64 <- box3d_iou(a[i], b[j])[0]; a (n,8,3), b (m,8,3) float32 corners
extern "C" __attribute__((visibility("default")))
int iou3d_corners_iou3d(int n, const float *a, int m, const float *b, double *iou, void *stream) {
  if (n < 0 || m < 0) return (int)hipErrorInvalidValue;
  const long long pairs = (long long)n * m;
  if (pairs == 0) return 0;
  if (pairs > 0x7fffffffLL * 128) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(corners_iou_matrix_kernel, dim3((unsigned)((pairs + 127) / 128)), dim3(128), 0,
                     (hipStream_t)stream, n, m, a, b, iou);
  return (int)hipGetLastError();
}

// ovmax (nd) f64, jmax (nd) i32 <- best ground-truth box of det d among gt[gt_begin[d] ..
// gt_begin[d]+gt_count[d]); (-inf, -1) when gt_count[d] == 0
extern "C" __attribute__((visibility("default")))
int iou3d_corners_best_match(int nd, const float *det, const int *gt_begin, const int *gt_count,
                             const float *gt, double *ovmax, int *jmax, void *stream) {
  if (nd < 0) return (int)hipErrorInvalidValue;
  if (nd == 0) return 0;
  hipLaunchKernelGGL(corners_best_match_kernel, dim3((nd + 127) / 128), dim3(128), 0,
                     (hipStream_t)stream, nd, det, gt_begin, gt_count, gt, ovmax, jmax);
  return (int)hipGetLastError();
}
