/* oracle/lhs_oracle.c -- CPU restatement of the pseudo-label "lower-half suppression" NMS.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): the product path never calls this.
 *
 * Follows the reference
 *   utils/nms.py:168-214            lhs_3d_faster_samecls (greedy NMS that re-admits the upper half,
 *                                   by score, of every suppressed set; same-class pairs only)
 *   models/loss_helper_unlabeled.py:447-487   how the boxes are formed: axis-aligned bounds of the
 *                                   8 corners of get_3d_box (utils/box_util.py:335-358, roty :326)
 *                                   in the "upright camera" frame (models/ap_helper.py:28-35),
 *                                   stored as float32, then everything in float64
 * Pinned by tests/golden/lhs_nms_ref.npz, generated with the reference's own utils/nms.py and
 * utils/box_util.py imported in the build container (tests/golden/make_lhs_golden.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* box i: centre (depth frame x,y,z; float32 as the network produced it), size (l,w,h) and heading
 * angle in float64 (the reference decodes them with numpy: float64 mean sizes + float32
 * residuals, loss_helper_unlabeled.py:458-463) -> float32 [x1,y1,z1,x2,y2,z2] in the camera frame
 * (X = x, Y = -z, Z = y); loss_helper_unlabeled.py:456-473 */
void lhso_camera_aabb(int n, const float *center, const double *size, const double *heading,
                      float *aabb) {
  static const double sx[8] = {1, 1, -1, -1, 1, 1, -1, -1};
  static const double sy[8] = {1, 1, 1, 1, -1, -1, -1, -1};
  static const double sz[8] = {1, -1, -1, 1, 1, -1, -1, 1};
  for (int i = 0; i < n; ++i) {
    const double cx = center[i * 3 + 0], cy = -(double)center[i * 3 + 2], cz = center[i * 3 + 1];
    const double l = size[i * 3 + 0], w = size[i * 3 + 1], h = size[i * 3 + 2];
    const double c = cos(heading[i]), s = sin(heading[i]);
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int k = 0; k < 8; ++k) {
      const double x = sx[k] * l / 2, y = sy[k] * h / 2, z = sz[k] * w / 2;
      /* roty: [[c,0,s],[0,1,0],[-s,0,c]] (box_util.py:326-332) */
      const float px = (float)((c * x + 0.0 * y + s * z) + cx);
      const float py = (float)((0.0 * x + 1.0 * y + 0.0 * z) + cy);
      const float pz = (float)((-s * x + 0.0 * y + c * z) + cz);
      if (k == 0 || px < lo[0]) lo[0] = px;
      if (k == 0 || py < lo[1]) lo[1] = py;
      if (k == 0 || pz < lo[2]) lo[2] = pz;
      if (k == 0 || px > hi[0]) hi[0] = px;
      if (k == 0 || py > hi[1]) hi[1] = py;
      if (k == 0 || pz > hi[2]) hi[2] = pz;
    }
    for (int d = 0; d < 3; ++d) { aabb[i * 6 + d] = lo[d]; aabb[i * 6 + 3 + d] = hi[d]; }
  }
}

/* nms.py:168-214 for one scene.  aabb (n,6) float32, score (n) float32 (already the float32
 * product pos_obj*iou), cls (n).  picked[i] = 1 for every index the reference appends to `pick`.
 * Ties in the ascending argsort are broken by index (numpy's quicksort leaves them unspecified). */
static void nms_core(int n, const float *aabb, const float *score, const long long *cls,
                     double thresh, int old_type, int same_class, int readmit, double area_eps,
                     int *picked);

void lhso_nms_samecls(int n, const float *aabb, const float *score, const long long *cls,
                      double thresh, int old_type, int *picked) {
  nms_core(n, aabb, score, cls, thresh, old_type, 1, 1, 1e-8, picked);
}

/* utils/nms.py:77-116 nms_3d_faster (same_class = 0) and :118-166 nms_3d_faster_samecls
 * (same_class = 1): the plain greedy loop (nothing re-admitted, areas without the 1e-8). */
void lhso_nms3d_aabb(int n, const float *aabb, const float *score, const long long *cls,
                     double thresh, int old_type, int same_class, int *picked) {
  nms_core(n, aabb, score, cls, thresh, old_type, same_class, 0, 0.0, picked);
}

static void nms_core(int n, const float *aabb, const float *score, const long long *cls,
                     double thresh, int old_type, int same_class, int readmit, double area_eps,
                     int *picked) {
  int *order = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int *sup = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  double *area = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    picked[i] = 0;
    order[i] = i;
    area[i] = ((double)aabb[i * 6 + 3] - aabb[i * 6 + 0]) * ((double)aabb[i * 6 + 4] - aabb[i * 6 + 1]) *
                  ((double)aabb[i * 6 + 5] - aabb[i * 6 + 2]) + area_eps;
  }
  for (int a = 1; a < n; ++a) { /* stable insertion sort, ascending score */
    const int v = order[a];
    int p = a - 1;
    while (p >= 0 && score[order[p]] > score[v]) { order[p + 1] = order[p]; --p; }
    order[p + 1] = v;
  }
  int count = n;
  while (count > 0) {
    const int i = order[count - 1];
    picked[i] = 1;
    int ns = 0;
    for (int t = 0; t < count - 1; ++t) {
      const int j = order[t];
      double xx1 = aabb[i * 6 + 0] > aabb[j * 6 + 0] ? aabb[i * 6 + 0] : aabb[j * 6 + 0];
      double yy1 = aabb[i * 6 + 1] > aabb[j * 6 + 1] ? aabb[i * 6 + 1] : aabb[j * 6 + 1];
      double zz1 = aabb[i * 6 + 2] > aabb[j * 6 + 2] ? aabb[i * 6 + 2] : aabb[j * 6 + 2];
      double xx2 = aabb[i * 6 + 3] < aabb[j * 6 + 3] ? aabb[i * 6 + 3] : aabb[j * 6 + 3];
      double yy2 = aabb[i * 6 + 4] < aabb[j * 6 + 4] ? aabb[i * 6 + 4] : aabb[j * 6 + 4];
      double zz2 = aabb[i * 6 + 5] < aabb[j * 6 + 5] ? aabb[i * 6 + 5] : aabb[j * 6 + 5];
      const double l = xx2 - xx1 > 0 ? xx2 - xx1 : 0, w = yy2 - yy1 > 0 ? yy2 - yy1 : 0,
                   h = zz2 - zz1 > 0 ? zz2 - zz1 : 0;
      double o;
      if (old_type) {
        o = (l * w * h) / area[j];
      } else {
        const double inter = l * w * h;
        o = inter / (area[i] + area[j] - inter);
      }
      if (same_class) o = o * (cls[i] == cls[j] ? 1.0 : 0.0);
      if (o > thresh) sup[ns++] = t;
    }
    if (readmit)
      for (int c = 0; c < ns / 2; ++c) picked[order[sup[ns - c - 1]]] = 1;
    /* delete position count-1 and every suppressed position */
    int w = 0, s = 0;
    for (int t = 0; t < count - 1; ++t) {
      if (s < ns && sup[s] == t) { ++s; continue; }
      order[w++] = order[t];
    }
    count = w;
  }
  free(order); free(sup); free(area);
}
