// 3dioumatch_amd/csrc/lhs_nms.hip -- the pseudo-label filter's NMS on the device (gfx950).
//
// What it replaces: the per-scene Python/numpy loop of get_pseudo_labels
// (models/loss_helper_unlabeled.py:447-487): device->host copies of the 64 best teacher
// predictions, get_3d_box per box (utils/box_util.py:335-358), axis-aligned bounds in the camera
// frame, and lhs_3d_faster_samecls (utils/nms.py:168-214) -- a greedy same-class NMS that
// re-admits the upper half (by score) of every suppressed set.  The reference runs it on the CPU
// in every semi-supervised step.
//
// One wavefront per scene, lane = box (n <= 64 = MAX_NUM_OBJ).  Lanes are put in ascending score
// order (rank by 64 LDS compares), the greedy loop then works on 64-bit lane masks: the winner is
// the highest remaining lane, the suppressed set is a ballot, "the upper half of it" is a popcount
// on the ballot above each lane.  Arithmetic as the reference: float32 corners from float64
// decoding, float64 areas / overlaps.
#include "common.h"

namespace {

struct Aabb { float x1, y1, z1, x2, y2, z2; };

__device__ __forceinline__ Aabb camera_aabb(const float *c, const double *sz, double heading) {
  const double sx[8] = {1, 1, -1, -1, 1, 1, -1, -1};
  const double sy[8] = {1, 1, 1, 1, -1, -1, -1, -1};
  const double sz8[8] = {1, -1, -1, 1, 1, -1, -1, 1};
  const double cx = c[0], cy = -(double)c[2], cz = c[1];
  const double l = sz[0], w = sz[1], h = sz[2];
  const double co = cos(heading), si = sin(heading);
  Aabb o = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double x = sx[k] * l / 2, y = sy[k] * h / 2, z = sz8[k] * w / 2;
    const float px = (float)((co * x + 0.0 * y + si * z) + cx);
    const float py = (float)((0.0 * x + 1.0 * y + 0.0 * z) + cy);
    const float pz = (float)((-si * x + 0.0 * y + co * z) + cz);
    if (k == 0 || px < o.x1) o.x1 = px;
    if (k == 0 || py < o.y1) o.y1 = py;
    if (k == 0 || pz < o.z1) o.z1 = pz;
    if (k == 0 || px > o.x2) o.x2 = px;
    if (k == 0 || py > o.y2) o.y2 = py;
    if (k == 0 || pz > o.z2) o.z2 = pz;
  }
  return o;
}

__global__ void __launch_bounds__(64)
lhs_nms_kernel(int n, const float *__restrict__ center, const double *__restrict__ size,
               const double *__restrict__ heading, const float *__restrict__ score,
               const long long *__restrict__ cls, double thresh, int old_type, int same_class,
               int readmit, double area_eps, int *__restrict__ picked) {
  __shared__ float s_score[64];
  __shared__ Aabb s_box[64];
  __shared__ double s_area[64];
  __shared__ long long s_cls[64];
  __shared__ int s_orig[64];
  const int scene = blockIdx.x, lane = threadIdx.x;
  const size_t base = (size_t)scene * n;
  const bool live = lane < n;
  Aabb mine = {0, 0, 0, 0, 0, 0};
  float my_score = 0.f;
  long long my_cls = -1;
  if (live) {
    mine = camera_aabb(center + (base + lane) * 3, size + (base + lane) * 3, heading[base + lane]);
    my_score = score[base + lane];
    my_cls = cls[base + lane];
  }
  s_score[lane] = my_score;
  __syncthreads();
  // ascending argsort, ties by index (numpy leaves them unspecified)
  int rank = 0;
  if (live) {
    for (int k = 0; k < n; ++k) {
      const float sk = s_score[k];
      rank += (sk < my_score || (sk == my_score && k < lane)) ? 1 : 0;
    }
    s_box[rank] = mine;
    s_area[rank] = ((double)mine.x2 - mine.x1) * ((double)mine.y2 - mine.y1) *
                       ((double)mine.z2 - mine.z1) + area_eps;
    s_cls[rank] = my_cls;
    s_orig[rank] = lane;
  }
  __syncthreads();
  // from here on lane r IS the box of rank r
  const Aabb b = live ? s_box[lane] : mine;
  const double area = live ? s_area[lane] : 1.0;
  const long long c = live ? s_cls[lane] : -1;
  unsigned long long remaining = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
  unsigned long long pick = 0ull;
  while (remaining) {
    const int i = 63 - __builtin_clzll(remaining);
    pick |= 1ull << i;
    const Aabb bi = s_box[i];
    const double area_i = s_area[i];
    const long long ci = s_cls[i];
    bool suppressed = false;
    if (((remaining >> lane) & 1ull) && lane != i) {
      const double xx1 = bi.x1 > b.x1 ? bi.x1 : b.x1, yy1 = bi.y1 > b.y1 ? bi.y1 : b.y1,
                   zz1 = bi.z1 > b.z1 ? bi.z1 : b.z1;
      const double xx2 = bi.x2 < b.x2 ? bi.x2 : b.x2, yy2 = bi.y2 < b.y2 ? bi.y2 : b.y2,
                   zz2 = bi.z2 < b.z2 ? bi.z2 : b.z2;
      const double l = xx2 - xx1 > 0 ? xx2 - xx1 : 0, w = yy2 - yy1 > 0 ? yy2 - yy1 : 0,
                   h = zz2 - zz1 > 0 ? zz2 - zz1 : 0;
      double o;
      if (old_type) {
        o = (l * w * h) / area;
      } else {
        const double inter = l * w * h;
        o = inter / (area_i + area - inter);
      }
      if (same_class) o = o * (ci == c ? 1.0 : 0.0);
      suppressed = o > thresh;
    }
    const unsigned long long sup = __ballot(suppressed);
    const int keep_n = readmit ? __popcll(sup) / 2 : 0;  // LHS: the upper half by score returns
    const unsigned long long above = lane >= 63 ? 0ull : (sup >> (lane + 1));
    const unsigned long long readmit = __ballot(suppressed && __popcll(above) < keep_n);
    pick |= readmit;
    remaining &= ~(sup | (1ull << i));
  }
  if (live) picked[base + s_orig[lane]] = (int)((pick >> lane) & 1ull);
}

// The same greedy loop for 64 < n <= T (evaluation: every proposal of a scene takes part,
// models/ap_helper.py:101-215): one T-lane workgroup per scene (T = 256, or 1024 for scenes with
// up to 1024 proposals), lane = box in ascending score order, the alive / suppressed sets are
// T/64 ballot words in LDS.
template <int T>
__global__ void __launch_bounds__(T)
nms_aabb_block_kernel(int n, const float *__restrict__ center, const double *__restrict__ size,
                      const double *__restrict__ heading, const float *__restrict__ score,
                      const long long *__restrict__ cls, double thresh, int old_type,
                      int same_class, int readmit, double area_eps, int *__restrict__ picked) {
  __shared__ float s_score[T];
  __shared__ Aabb s_box[T];
  __shared__ double s_area[T];
  __shared__ long long s_cls[T];
  __shared__ int s_orig[T];
  __shared__ unsigned long long s_alive[T / 64], s_sup[T / 64];
  const int scene = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t base = (size_t)scene * n;
  const bool live = tid < n;
  Aabb mine = {0, 0, 0, 0, 0, 0};
  float my_score = 0.f;
  long long my_cls = -1;
  if (live) {
    mine = camera_aabb(center + (base + tid) * 3, size + (base + tid) * 3, heading[base + tid]);
    my_score = score[base + tid];
    my_cls = cls[base + tid];
  }
  s_score[tid] = my_score;
  __syncthreads();
  if (live) {
    int rank = 0;
    for (int k = 0; k < n; ++k) {
      const float sk = s_score[k];
      rank += (sk < my_score || (sk == my_score && k < tid)) ? 1 : 0;
    }
    s_box[rank] = mine;
    s_area[rank] = ((double)mine.x2 - mine.x1) * ((double)mine.y2 - mine.y1) *
                       ((double)mine.z2 - mine.z1) + area_eps;
    s_cls[rank] = my_cls;
    s_orig[rank] = tid;
  }
  __syncthreads();
  const Aabb b = live ? s_box[tid] : mine;   // lane r is now the box of rank r
  const double area = live ? s_area[tid] : 1.0;
  const long long c = live ? s_cls[tid] : -1;
  bool alive = live, pick = false;
  for (;;) {
    const unsigned long long am = __ballot(alive);
    if (lane == 0) s_alive[w] = am;
    __syncthreads();
    int i = -1;
#pragma unroll
    for (int q = T / 64 - 1; q >= 0; --q)
      if (i < 0 && s_alive[q]) i = q * 64 + 63 - __builtin_clzll(s_alive[q]);
    if (i < 0) break;  // uniform: every lane read the same words
    const Aabb bi = s_box[i];
    const double area_i = s_area[i];
    const long long ci = s_cls[i];
    bool suppressed = false;
    if (alive && tid != i) {
      const double xx1 = bi.x1 > b.x1 ? bi.x1 : b.x1, yy1 = bi.y1 > b.y1 ? bi.y1 : b.y1,
                   zz1 = bi.z1 > b.z1 ? bi.z1 : b.z1;
      const double xx2 = bi.x2 < b.x2 ? bi.x2 : b.x2, yy2 = bi.y2 < b.y2 ? bi.y2 : b.y2,
                   zz2 = bi.z2 < b.z2 ? bi.z2 : b.z2;
      const double l = xx2 - xx1 > 0 ? xx2 - xx1 : 0, wd = yy2 - yy1 > 0 ? yy2 - yy1 : 0,
                   h = zz2 - zz1 > 0 ? zz2 - zz1 : 0;
      double o;
      if (old_type) {
        o = (l * wd * h) / area;
      } else {
        const double inter = l * wd * h;
        o = inter / (area_i + area - inter);
      }
      if (same_class) o = o * (ci == c ? 1.0 : 0.0);
      suppressed = o > thresh;
    }
    const unsigned long long sm = __ballot(suppressed);
    if (lane == 0) s_sup[w] = sm;
    __syncthreads();
    if (tid == i) { pick = true; alive = false; }
    if (suppressed) {
      if (readmit) {
        int total = 0, above = lane >= 63 ? 0 : __popcll(s_sup[w] >> (lane + 1));
#pragma unroll
        for (int q = 0; q < T / 64; ++q) {
          total += __popcll(s_sup[q]);
          if (q > w) above += __popcll(s_sup[q]);
        }
        if (above < total / 2) pick = true;
      }
      alive = false;
    }
    __syncthreads();  // s_alive / s_sup are rewritten by the next round
  }
  if (live) picked[base + s_orig[tid]] = pick ? 1 : 0;
}

}  // namespace

static int nms_aabb_launch(int scenes, int n, const float *center, const double *size,
                           const double *heading, const float *score, const long long *cls,
                           double thresh, int old_type, int same_class, int readmit,
                           double area_eps, int *picked, hipStream_t stream) {
  if (scenes <= 0 || n <= 0) return 0;
  if (n > 1024) return (int)hipErrorInvalidValue;
  if (n <= 64)
    hipLaunchKernelGGL(lhs_nms_kernel, dim3(scenes), dim3(64), 0, stream, n, center, size, heading,
                       score, cls, thresh, old_type, same_class, readmit, area_eps, picked);
  else if (n <= 256)
    hipLaunchKernelGGL(nms_aabb_block_kernel<256>, dim3(scenes), dim3(256), 0, stream, n, center,
                       size, heading, score, cls, thresh, old_type, same_class, readmit, area_eps,
                       picked);
  else
    hipLaunchKernelGGL(nms_aabb_block_kernel<1024>, dim3(scenes), dim3(1024), 0, stream, n, center,
                       size, heading, score, cls, thresh, old_type, same_class, readmit, area_eps,
                       picked);
  return pn2_launch_status();
}

// picked (scenes, n) int32 <- 1 for every box the reference's greedy axis-aligned 3-D NMS returns:
// nms_3d_faster (same_class = 0) / nms_3d_faster_samecls (same_class = 1), utils/nms.py:77-166,
// n <= 1024 (the evaluation path, models/ap_helper.py:170-203)
extern "C" __attribute__((visibility("default")))
int lhs_nms3d_aabb(int scenes, int n, const float *center, const double *size,
                   const double *heading, const float *score, const long long *cls, double thresh,
                   int old_type, int same_class, int *picked, void *stream) {
  return nms_aabb_launch(scenes, n, center, size, heading, score, cls, thresh, old_type,
                         same_class, 0, 0.0, picked, (hipStream_t)stream);
}

// picked (scenes, n) int32 <- 1 for every box lhs_3d_faster_samecls returns (utils/nms.py:168-214)
extern "C" __attribute__((visibility("default")))
int lhs_nms_samecls(int scenes, int n, const float *center, const double *size,
                    const double *heading, const float *score, const long long *cls, double thresh,
                    int old_type, int *picked, void *stream) {
  if (n > 64) return (int)hipErrorInvalidValue;  // MAX_NUM_OBJ = 64 (loss_helper_unlabeled.py:21)
  return nms_aabb_launch(scenes, n, center, size, heading, score, cls, thresh, old_type, 1, 1, 1e-8,
                         picked, (hipStream_t)stream);
}
