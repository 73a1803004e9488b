// 3dioumatch_amd/csrc/gridconv_front.hip -- front end of the IoU branch's GridConv.
//
// What it replaces (models/grid_conv_module.py:64-98, per call and twice per train step):
//   * the 4x4x4 grid inside every proposal box -- linspace / repeat / expand / cat, the scaling by
//     the box size, rot_gpu + torch.bmm, the two additions of the centre and the subtraction that
//     gives the relative coordinates (:64-83, :91): ~15 tensor kernels -> gridconv_points_kernel,
//     which also writes the relative coordinates straight into channels 0..2 of the (B, 3+C, K*64)
//     tensor the shared MLP reads;
//   * the inverse-distance weights of the three nearest seeds -- sqrt, + 1e-8, reciprocal, sum,
//     divide (:94-98; the same five operations in PointnetFPModule.forward,
//     pointnet2_modules.py:395-398) -> three_nn_weights_kernel.
// Arithmetic: every operation of the tensor formulation, in its order, each rounded to fp32
// (-ffp-contract=off): unit * size, lx*cos + ly*sin, ly*cos - lx*sin, + centre, - centre;
// 1 / (sqrt(d2) + 1e-8), (w0 + w1) + w2, w / sum.  The unit grid (linspace(-1, 1, 4), x slowest /
// z fastest) is passed in, as computed by torch once.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
gridconv_points_kernel(int total, int k, int ctot, const float *__restrict__ unit,
                       const float *__restrict__ center, const float *__restrict__ size,
                       const float *__restrict__ heading, float *__restrict__ whole,
                       float *__restrict__ feats) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // (cloud * k + proposal) * 64 + grid point
  if (t >= total) return;
  const int g = t & 63, box = t >> 6, b = box / k, kk = box - b * k;
  const float lx = __fmul_rn(unit[g * 3 + 0], size[box * 3 + 0]);
  const float ly = __fmul_rn(unit[g * 3 + 1], size[box * 3 + 1]);
  const float lz = __fmul_rn(unit[g * 3 + 2], size[box * 3 + 2]);
  const float h = heading[box];
  const float c = cosf(h), s = sinf(h);
  const float cx = center[box * 3 + 0], cy = center[box * 3 + 1], cz = center[box * 3 + 2];
  const float wx = __fadd_rn(__fadd_rn(__fmul_rn(lx, c), __fmul_rn(ly, s)), cx);
  const float wy = __fadd_rn(__fsub_rn(__fmul_rn(ly, c), __fmul_rn(lx, s)), cy);
  const float wz = __fadd_rn(lz, cz);
  float *w = whole + (size_t)t * 3;
  w[0] = wx; w[1] = wy; w[2] = wz;
  const size_t cols = (size_t)k * 64;
  float *f = feats + (size_t)b * ctot * cols + (size_t)kk * 64 + g;
  f[0] = __fsub_rn(wx, cx);
  f[cols] = __fsub_rn(wy, cy);
  f[2 * cols] = __fsub_rn(wz, cz);
}

__global__ void __launch_bounds__(256)
three_nn_weights_kernel(long long n, const float *__restrict__ dist2, float *__restrict__ weight) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const float *d = dist2 + t * 3;
  const float w0 = __fdiv_rn(1.0f, __fadd_rn(sqrtf(d[0]), 1e-8f));
  const float w1 = __fdiv_rn(1.0f, __fadd_rn(sqrtf(d[1]), 1e-8f));
  const float w2 = __fdiv_rn(1.0f, __fadd_rn(sqrtf(d[2]), 1e-8f));
  const float sum = __fadd_rn(__fadd_rn(w0, w1), w2);
  float *o = weight + t * 3;
  o[0] = __fdiv_rn(w0, sum);
  o[1] = __fdiv_rn(w1, sum);
  o[2] = __fdiv_rn(w2, sum);
}

// ---- decoded boxes of the proposals + one jittered copy of each --------------------------------
// What it replaces: VoteNet.calculate_bbox + the jitter of forward_with_pred_jitter
// (models/votenet_iou_branch.py:111-137, :157-172) in the TRAINING forward, where nothing flows back
// through these tensors (they feed the detached IoU branch and the IoU labels): two argmax, two
// gathers, index_select, the size / angle arithmetic, where, two randn-scaled offsets, clamp, clone
// and three cats -- ~25 tensor kernels.  One lane per proposal; every operation of the tensor
// formulation in its order, each rounded to fp32; the two noise tensors are drawn by torch (same
// generator stream as the reference).  arg-max = first maximum.
__global__ void __launch_bounds__(256)
bbox_jitter_kernel(int total, int k, int ns, int nh, float angle_per_class,
                   const float *__restrict__ center, const float *__restrict__ size_scores,
                   const float *__restrict__ size_residuals,
                   const float *__restrict__ heading_scores,
                   const float *__restrict__ heading_residuals, const float *__restrict__ mean_size,
                   const float *__restrict__ noise_c, const float *__restrict__ noise_s,
                   float *__restrict__ size, float *__restrict__ heading,
                   float *__restrict__ all_center, float *__restrict__ all_size,
                   float *__restrict__ all_heading, float *__restrict__ jitter_size2) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // cloud * k + proposal
  if (t >= total) return;
  const int b = t / k, kk = t - b * k;
  const float *ss = size_scores + (size_t)t * ns;
  int sc = 0;
  float best = ss[0];
  for (int q = 1; q < ns; ++q)
    if (ss[q] > best) { best = ss[q]; sc = q; }
  float sz[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float v = __fdiv_rn(__fadd_rn(mean_size[sc * 3 + d], size_residuals[((size_t)t * ns + sc) * 3 + d]), 2.0f);
    sz[d] = v < 0.f ? 1e-6f : v;
  }
  float hd = 0.f;
  if (nh > 1) {
    const float *hs = heading_scores + (size_t)t * nh;
    int hc = 0;
    float hb = hs[0];
    for (int q = 1; q < nh; ++q)
      if (hs[q] > hb) { hb = hs[q]; hc = q; }
    hd = __fadd_rn(__fmul_rn((float)hc, angle_per_class), heading_residuals[(size_t)t * nh + hc]);
    const float kPi = 3.14159265358979323846f, kTwoPi = 6.28318530717958647692f;
    hd = __fsub_rn(hd, __fmul_rn(hd > kPi ? 1.0f : 0.0f, kTwoPi));
  }
  const size_t o1 = ((size_t)b * 2 * k + kk) * 3, o2 = ((size_t)b * 2 * k + k + kk) * 3;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float c = center[(size_t)t * 3 + d];
    size[(size_t)t * 3 + d] = sz[d];
    const float cj = __fadd_rn(c, __fmul_rn(__fmul_rn(sz[d], noise_c[(size_t)t * 3 + d]), 0.3f));
    float sj = __fadd_rn(sz[d], __fmul_rn(__fmul_rn(sz[d], noise_s[(size_t)t * 3 + d]), 0.3f));
    sj = sj < 1e-8f ? 1e-8f : sj;
    all_center[o1 + d] = c;
    all_center[o2 + d] = cj;
    all_size[o1 + d] = sz[d];
    all_size[o2 + d] = sj;
    jitter_size2[(size_t)t * 3 + d] = __fmul_rn(sj, 2.0f);
  }
  heading[t] = hd;
  all_heading[(size_t)b * 2 * k + kk] = hd;
  all_heading[(size_t)b * 2 * k + k + kk] = hd;
}

}  // namespace

// proposals of b clouds x k: decoded (size, heading), the (b,2k,*) tensors [predicted | jittered]
// the IoU branch consumes, and 2 * jittered size (the `jitter_size` entry of the end points)
PN2_API int votenet_bbox_jitter(int b, int k, int ns, int nh, const float *center,
                                const float *size_scores, const float *size_residuals,
                                const float *heading_scores, const float *heading_residuals,
                                const float *mean_size, const float *noise_c, const float *noise_s,
                                float *size, float *heading, float *all_center, float *all_size,
                                float *all_heading, float *jitter_size2, void *stream_) {
  if (b <= 0 || k <= 0) return 0;
  if (ns < 1 || nh < 1 || !center || !size_scores || !size_residuals || !mean_size || !noise_c ||
      !noise_s || !size || !heading || !all_center || !all_size || !all_heading || !jitter_size2 ||
      (nh > 1 && (!heading_scores || !heading_residuals)))
    return (int)hipErrorInvalidValue;
  const int total = b * k;
  const float angle_per_class = (float)(2.0 * 3.14159265358979323846 / (double)nh);
  hipLaunchKernelGGL(bbox_jitter_kernel, dim3(pn2_ceil_div(total, 256)), dim3(256), 0,
                     (hipStream_t)stream_, total, k, ns, nh, angle_per_class, center, size_scores,
                     size_residuals, heading_scores, heading_residuals, mean_size, noise_c, noise_s,
                     size, heading, all_center, all_size, all_heading, jitter_size2);
  return pn2_launch_status();
}

// center, size (b,k,3), heading (b,k), unit (64,3) -> whole (b,k*64,3) and channels 0..2 of
// feats (b,ctot,k*64)
PN2_API int votenet_gridconv_points(int b, int k, int ctot, const float *unit, const float *center,
                                    const float *size, const float *heading, float *whole,
                                    float *feats, void *stream_) {
  if (b <= 0 || k <= 0) return 0;
  if (ctot < 3 || !unit || !center || !size || !heading || !whole || !feats) return (int)hipErrorInvalidValue;
  const long long total = (long long)b * k * 64;
  if (total > 0x7fffffffll) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gridconv_points_kernel, dim3(pn2_ceil_div(total, 256)), dim3(256), 0,
                     (hipStream_t)stream_, (int)total, k, ctot, unit, center, size, heading, whole, feats);
  return pn2_launch_status();
}

// dist2 (n,3) squared distances of three_nn -> weight (n,3) = normalised 1 / (sqrt(d2) + 1e-8)
PN2_API int pn2_three_nn_weights(long long n, const float *dist2, float *weight, void *stream_) {
  if (n <= 0) return 0;
  if (!dist2 || !weight) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(three_nn_weights_kernel, dim3(pn2_ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream_, n, dist2, weight);
  return pn2_launch_status();
}
