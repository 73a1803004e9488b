// 3dioumatch_amd/csrc/feat_norm.hip -- unit-length vote features.
//
// What it replaces: models/votenet_iou_branch.py:103-104,
//     features_norm = torch.norm(features, p=2, dim=1); features = features.div(features_norm.unsqueeze(1))
// and its autograd backward (about thirty element-wise / reduction kernels over a (B, 256, 1024)
// tensor in the training step).  One kernel each way:
//     forward : norm[b][j] = sqrt(sum_c x[b][c][j]^2),  y = x / norm
//     backward: dx = (dy - y * sum_c(dy * y)) / norm
// A workgroup owns 64 columns of one cloud; its 4 waves split the channels, partial sums meet in
// LDS.  No epsilon, as in the reference (a zero column gives the same NaNs).
#include "common.h"

namespace {

// sum over channels of f(channel value(s)) for column j, channels split over the 4 waves
template <bool GRAD>
__global__ void __launch_bounds__(256)
channel_normalize_kernel(int c, int n, const float *__restrict__ x, const float *__restrict__ dy,
                         float *__restrict__ norm, float *__restrict__ out) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane, b = blockIdx.y;
  const bool live = j < n;
  const size_t base = (size_t)b * c * n + (live ? j : 0);
  const int c_lo = (int)((long long)c * wave / 4), c_hi = (int)((long long)c * (wave + 1) / 4);
  float acc = 0.f;
  if (live) {
    // eight rows in flight per lane (128 workgroups x 4 waves with one 256-byte load each in
    // flight left the kernel at 0.6 TB/s); the sum keeps its channel order
    int ch = c_lo;
    for (; ch + 8 <= c_hi; ch += 8) {
      float v[8], g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = x[base + (size_t)(ch + u) * n];  // forward: x; backward: y
        g[u] = GRAD ? dy[base + (size_t)(ch + u) * n] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += GRAD ? g[u] * v[u] : v[u] * v[u];
    }
    for (; ch < c_hi; ++ch) {
      const float v = x[base + (size_t)ch * n];
      acc += GRAD ? dy[base + (size_t)ch * n] * v : v * v;
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  const float total = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  if (!live) return;
  if (!GRAD) {
    const float len = sqrtf(total);
    if (wave == 0) norm[(size_t)b * n + j] = len;
#pragma unroll 8
    for (int ch = c_lo; ch < c_hi; ++ch) out[base + (size_t)ch * n] = x[base + (size_t)ch * n] / len;
  } else {
    const float len = norm[(size_t)b * n + j];
#pragma unroll 8
    for (int ch = c_lo; ch < c_hi; ++ch) {
      const size_t o = base + (size_t)ch * n;
      out[o] = (dy[o] - x[o] * total) / len;
    }
  }
}

}  // namespace

// x (b,c,n) -> y (b,c,n) = x / ||x||_2 over c, norm (b,n)
PN2_API int votenet_channel_normalize(int b, int c, int n, const float *x, float *y, float *norm,
                                      void *stream_) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(channel_normalize_kernel<false>, dim3(pn2_ceil_div(n, 64), b), dim3(256), 0,
                     (hipStream_t)stream_, c, n, x, nullptr, norm, y);
  return pn2_launch_status();
}

// dx (b,c,n) from dy, the forward's outputs y and norm
PN2_API int votenet_channel_normalize_grad(int b, int c, int n, const float *y, const float *norm,
                                           const float *dy, float *dx, void *stream_) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(channel_normalize_kernel<true>, dim3(pn2_ceil_div(n, 64), b), dim3(256), 0,
                     (hipStream_t)stream_, c, n, y, dy, const_cast<float *>(norm), dx);
  return pn2_launch_status();
}
