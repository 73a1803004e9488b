/* oracle/evaldet_oracle.c -- CPU restatement of the oriented-box IoU of the AP evaluation.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): the product path never calls this.
 *
 * Follows the reference
 *   utils/box_util.py:23-69     polygon_clip (Sutherland-Hodgman, strict inside test)
 *   utils/box_util.py:77-88     convex_hull_intersection (area of the clipped polygon)
 *   utils/box_util.py:91-96     box3d_vol
 *   utils/box_util.py:112-137   box3d_iou (footprints = vertices 3,2,1,0 in the x-z plane; vertical
 *                               extent from vertices 0 and 4)
 *   utils/eval_det.py:128-141   per-detection ovmax / jmax with a strict `>` update, on
 *                               float64 copies of the float32 corner arrays
 * The reference takes the clipped polygon's area from scipy.spatial.ConvexHull(...).volume; the
 * polygon is convex, so this is its shoelace area, which is what is computed here (and 0 where the
 * polygon has fewer than 3 vertices; the reference raises QhullError there).
 * Pinned by tests/golden/evaldet_ref.npz, generated with the reference's own utils/box_util.py,
 * utils/eval_det.py and models/ap_helper.py imported in the build container
 * (tests/golden/make_evaldet_golden.py).
 */
#include <math.h>

typedef struct { double x, y; } evo_p2;

static int evo_inside(evo_p2 cp1, evo_p2 cp2, evo_p2 p) {
  return (cp2.x - cp1.x) * (p.y - cp1.y) > (cp2.y - cp1.y) * (p.x - cp1.x);
}

static evo_p2 evo_intersection(evo_p2 cp1, evo_p2 cp2, evo_p2 s, evo_p2 e) {
  const double dc0 = cp1.x - cp2.x, dc1 = cp1.y - cp2.y;
  const double dp0 = s.x - e.x, dp1 = s.y - e.y;
  const double n1 = cp1.x * cp2.y - cp1.y * cp2.x;
  const double n2 = s.x * e.y - s.y * e.x;
  const double n3 = 1.0 / (dc0 * dp1 - dc1 * dp0);
  evo_p2 q;
  q.x = (n1 * dp0 - n2 * dc0) * n3;
  q.y = (n1 * dp1 - n2 * dc1) * n3;
  return q;
}

/* polygon_clip; out must hold 16 vertices; returns the vertex count (0 = None) */
static int evo_polygon_clip(const evo_p2 *subject, int ns, const evo_p2 *clip, int nc, evo_p2 *out) {
  evo_p2 in[16];
  int n_out = ns;
  for (int i = 0; i < ns; ++i) out[i] = subject[i];
  evo_p2 cp1 = clip[nc - 1];
  for (int c = 0; c < nc; ++c) {
    const evo_p2 cp2 = clip[c];
    const int n_in = n_out;
    for (int i = 0; i < n_in; ++i) in[i] = out[i];
    n_out = 0;
    evo_p2 s = in[n_in - 1];
    for (int i = 0; i < n_in; ++i) {
      const evo_p2 e = in[i];
      if (evo_inside(cp1, cp2, e)) {
        if (!evo_inside(cp1, cp2, s) && n_out < 16) out[n_out++] = evo_intersection(cp1, cp2, s, e);
        if (n_out < 16) out[n_out++] = e;
      } else if (evo_inside(cp1, cp2, s)) {
        if (n_out < 16) out[n_out++] = evo_intersection(cp1, cp2, s, e);
      }
      s = e;
    }
    cp1 = cp2;
    if (n_out == 0) return 0;
  }
  return n_out;
}

static double evo_edge(const float *c, int i, int j) {
  double acc = 0.0;
  for (int k = 0; k < 3; ++k) {
    const double d = (double)c[i * 3 + k] - (double)c[j * 3 + k];
    acc = (k == 0) ? d * d : acc + d * d;
  }
  return sqrt(acc);
}

/* box3d_iou(corners1, corners2)[0]; corners (8,3) float32 */
double evo_box3d_iou(const float *c1, const float *c2) {
  evo_p2 r1[4], r2[4], inter[16];
  for (int k = 0; k < 4; ++k) {
    r1[k].x = c1[(3 - k) * 3 + 0]; r1[k].y = c1[(3 - k) * 3 + 2];
    r2[k].x = c2[(3 - k) * 3 + 0]; r2[k].y = c2[(3 - k) * 3 + 2];
  }
  const int n = evo_polygon_clip(r1, 4, r2, 4, inter);
  double inter_area = 0.0;
  if (n >= 3) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
      const evo_p2 a = inter[i], b = inter[(i + 1) % n];
      acc += a.x * b.y - a.y * b.x;
    }
    inter_area = 0.5 * fabs(acc);
  }
  const double y1 = c1[1], y2 = c2[1], y1b = c1[4 * 3 + 1], y2b = c2[4 * 3 + 1];
  const double ymax = y1 < y2 ? y1 : y2;     /* min(corners1[0,1], corners2[0,1]) */
  const double ymin = y1b > y2b ? y1b : y2b; /* max(corners1[4,1], corners2[4,1]) */
  const double dy = ymax - ymin;
  const double inter_vol = inter_area * (dy > 0.0 ? dy : 0.0);
  const double vol1 = evo_edge(c1, 0, 1) * evo_edge(c1, 1, 2) * evo_edge(c1, 0, 4);
  const double vol2 = evo_edge(c2, 0, 1) * evo_edge(c2, 1, 2) * evo_edge(c2, 0, 4);
  return inter_vol / (vol1 + vol2 - inter_vol);
}

/* iou (n,m) <- box3d_iou(a[i], b[j]) */
void evo_iou_matrix(int n, const float *a, int m, const float *b, double *iou) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) iou[(long)i * m + j] = evo_box3d_iou(a + (long)i * 24, b + (long)j * 24);
}

/* eval_det.py:128-141 for every detection d over gt[gt_begin[d] .. +gt_count[d]) */
void evo_best_match(int nd, const float *det, const int *gt_begin, const int *gt_count,
                    const float *gt, double *ovmax, int *jmax) {
  for (int d = 0; d < nd; ++d) {
    double best = -INFINITY;
    int bj = -1;
    for (int j = 0; j < gt_count[d]; ++j) {
      const double v = evo_box3d_iou(det + (long)d * 24, gt + (long)(gt_begin[d] + j) * 24);
      if (v > best) { best = v; bj = j; }
    }
    ovmax[d] = best;
    jmax[d] = bj;
  }
}
