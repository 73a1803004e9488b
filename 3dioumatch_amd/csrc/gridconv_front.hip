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

}  // namespace

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
