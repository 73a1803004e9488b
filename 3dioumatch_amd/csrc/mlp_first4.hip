// 3dioumatch_amd/csrc/mlp_first4.hip -- weight gradient of a set-abstraction level's FIRST layer
// when its input has 4 channels (SA1: xyz + height, conv 4 -> 64).
//
// What it replaces: the backward-weight of that nn.Conv2d (pointnet2/pytorch_utils.py:70-124)
// behind BatchNorm2d + ReLU, i.e. dW = sum_n dy[:, n] x[:, n]^T with
//     dy = a * (g - c1 - xhat * c2),   g = [y*sc + sh > 0] * dz,   xhat = (y - mu) * is,  y = W x.
// The general wgrad kernel streams the pair (y, dz): 2 x 64 rows x 1 M columns = 537 MB at config
// 2, the whole cost of the launch.  But y is a rank-4 function of x, so only the GATED term needs
// the big tensor:
//     dW[k][c] = a_k * ( G[k][c] - c1_k * X1[c] - c2_k * is_k * ( (W S)[k][c] - mu_k * X1[c] ) )
//     G[k][c] = sum_n g[k][n] x[c][n],   S = sum_n x x^T (4 x 4),   X1 = sum_n x
// G streams dz once (268 MB) next to x (17 MB) and recomputes y -- four FMAs -- for the ReLU
// gate; S and X1 are 14 sums over x alone.  Three launches: gated sums (one partial per
// workgroup), moments (double accumulators), combine.
#include "common.h"
#include "mlp_operand.h"

namespace {

constexpr int kF4Rows = 64;

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask, kWave);
  hi = __shfl_xor(hi, mask, kWave);
  return __hiloint2double(hi, lo);
}

// G partials: 8 waves, wave w owns rows 8w .. 8w+7, a lane owns 4 consecutive columns per step
constexpr int kF4Rpw = 8;  // rows per wave
__global__ void __launch_bounds__(512, 2)
first4_gated_kernel(int r, int per, const float *__restrict__ w, const float *__restrict__ x,
                    const float *__restrict__ dz, const float *__restrict__ scale,
                    const float *__restrict__ shift, float *__restrict__ part) {
  __shared__ float4 wrow[kF4Rows];
  __shared__ float2 gate[kF4Rows];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, b = blockIdx.y, slices = gridDim.x;
  if (tid < kF4Rows) {
    wrow[tid] = make_float4(w[tid * 4], w[tid * 4 + 1], w[tid * 4 + 2], w[tid * 4 + 3]);
    gate[tid] = make_float2(scale[tid], shift[tid]);
  }
  __syncthreads();
  const int r_lo = s * per, r_hi = r_lo + per < r ? r_lo + per : r;
  const float *xb = x + (size_t)b * 4 * r;
  const float *db = dz + (size_t)b * kF4Rows * r;
  float acc[kF4Rpw][4];
#pragma unroll
  for (int i = 0; i < kF4Rpw; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
  // few waves per SIMD (the 64 accumulators and two sets of loads): the loads of step s+1 are
  // issued before step s is reduced, so the HBM latency hides behind ~700 VALU instructions
  float4 xv[4], dv[kF4Rpw];
  auto fetch = [&](int c0) {
    const bool in = c0 < r_hi;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      xv[k] = in ? *reinterpret_cast<const float4 *>(xb + (size_t)k * r + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < kF4Rpw; ++i)
      dv[i] = in ? *reinterpret_cast<const float4 *>(db + (size_t)(wave * kF4Rpw + i) * r + c0)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  fetch(r_lo + lane * 4);
  for (int c0 = r_lo + lane * 4; c0 < r_hi; c0 += 256) {  // r, per: multiples of 4
    float xe[4][4], de[kF4Rpw][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { xe[k][0] = xv[k].x; xe[k][1] = xv[k].y; xe[k][2] = xv[k].z; xe[k][3] = xv[k].w; }
#pragma unroll
    for (int i = 0; i < kF4Rpw; ++i) { de[i][0] = dv[i].x; de[i][1] = dv[i].y; de[i][2] = dv[i].z; de[i][3] = dv[i].w; }
    fetch(c0 + 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < kF4Rpw; ++i) {
      const int row = wave * kF4Rpw + i;
      const float4 wr = wrow[row];
      const float2 gt = gate[row];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = lin4(wr, xe[0][e], xe[1][e], xe[2][e], xe[3][e]);
        const float g = __fmaf_rn(y, gt.x, gt.y) > 0.f ? de[i][e] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = __fmaf_rn(g, xe[k][e], acc[i][k]);
      }
    }
  }
  float *out = part + ((size_t)b * slices + s) * (kF4Rows * 4);
#pragma unroll
  for (int i = 0; i < kF4Rpw; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = acc[i][k];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
      if (lane == 0) out[(wave * kF4Rpw + i) * 4 + k] = v;
    }
}

// moments of x: 4 first and 10 second sums per workgroup, in double
__global__ void __launch_bounds__(256)
first4_moments_kernel(int r, long long total, const float *__restrict__ x, double *__restrict__ mpart) {
  __shared__ double red[4][14];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double m[14];
#pragma unroll
  for (int q = 0; q < 14; ++q) m[q] = 0.0;
  // column index over the whole batch: cloud = col / r
  for (long long col = (long long)blockIdx.x * 256 + tid; col < total; col += (long long)gridDim.x * 256) {
    const long long b = col / r, n = col - b * r;
    const float *p = x + (size_t)b * 4 * r + n;
    const double x0 = p[0], x1 = p[r], x2 = p[2 * (size_t)r], x3 = p[3 * (size_t)r];
    m[0] += x0; m[1] += x1; m[2] += x2; m[3] += x3;
    m[4] += x0 * x0; m[5] += x0 * x1; m[6] += x0 * x2; m[7] += x0 * x3;
    m[8] += x1 * x1; m[9] += x1 * x2; m[10] += x1 * x3;
    m[11] += x2 * x2; m[12] += x2 * x3; m[13] += x3 * x3;
  }
#pragma unroll
  for (int q = 0; q < 14; ++q) {
    double v = m[q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += shfl_xor_f64(v, off);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (tid < 14) mpart[(size_t)blockIdx.x * 14 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// dW from the reduced sums; one lane per (k, c).  gsum: the 256 gated sums already reduced over
// the workgroups; the 256 moment partials are reduced here (lane = partial, then a butterfly).
__global__ void __launch_bounds__(256)
first4_combine_kernel(const float *__restrict__ gsum, const double *__restrict__ mpart,
                      const float *__restrict__ w, const float *__restrict__ mean,
                      const float *__restrict__ invstd, const float *__restrict__ coef,
                      float *__restrict__ dw) {
  __shared__ double red[4][14];
  __shared__ double mom[14];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int q = 0; q < 14; ++q) {
    double v = mpart[(size_t)tid * 14 + q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += shfl_xor_f64(v, off);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (tid < 14) mom[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  __syncthreads();
  const int k = tid >> 2, c = tid & 3;
  const double g = gsum[tid];
  // S as a symmetric 4 x 4 from the 10 stored sums
  const int sidx[4][4] = {{4, 5, 6, 7}, {5, 8, 9, 10}, {6, 9, 11, 12}, {7, 10, 12, 13}};
  double ws = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) ws += (double)w[k * 4 + q] * mom[sidx[q][c]];
  const double a = coef[k * 3], c1 = coef[k * 3 + 1], c2 = coef[k * 3 + 2];
  const double x1 = mom[c];
  dw[tid] = (float)(a * (g - c1 * x1 - c2 * (double)invstd[k] * (ws - (double)mean[k] * x1)));
}

// BatchNorm coefficients of y = W x from the moments of x (no pass over y, which is never
// stored): mean_y = W mean_x, E[y^2] = w^T (S/N) w.  Same outputs and running-statistics update
// as mlp_bn_finalize_pairs.  One workgroup; lane = channel.
__global__ void __launch_bounds__(256)
first4_bn_kernel(const double *__restrict__ mpart, double count, const float *__restrict__ w,
                 const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                 float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
                 float *__restrict__ mean_out, float *__restrict__ invstd_out,
                 float *__restrict__ scale_out, float *__restrict__ shift_out) {
  __shared__ double red[4][14];
  __shared__ double mom[14];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int q = 0; q < 14; ++q) {
    double v = mpart[(size_t)tid * 14 + q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += shfl_xor_f64(v, off);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (tid < 14) mom[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  __syncthreads();
  if (tid >= kF4Rows) return;
  const int sidx[4][4] = {{4, 5, 6, 7}, {5, 8, 9, 10}, {6, 9, 11, 12}, {7, 10, 12, 13}};
  const double wk[4] = {w[tid * 4], w[tid * 4 + 1], w[tid * 4 + 2], w[tid * 4 + 3]};
  double mean = 0.0, ey2 = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    mean += wk[c] * mom[c];
#pragma unroll
    for (int d = 0; d < 4; ++d) ey2 += wk[c] * wk[d] * mom[sidx[c][d]];
  }
  mean /= count;
  double var = ey2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float fmean = (float)mean;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_out[tid] = fmean;
  invstd_out[tid] = invstd;
  const float sc = gamma[tid] * invstd;
  scale_out[tid] = sc;
  shift_out[tid] = beta[tid] - fmean * sc;
  if (running_mean != nullptr) {  // nn.BatchNorm: unbiased variance in the running estimate
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[tid] = (1.f - momentum) * running_mean[tid] + momentum * fmean;
    running_var[tid] = (1.f - momentum) * running_var[tid] + momentum * (float)unbiased;
  }
}

int first4_slices(int b, int r) {
  long long s = 512 / (b > 0 ? b : 1);  // two workgroups per CU, each streams many steps
  if (s < 1) s = 1;
  long long per = ((long long)r + s - 1) / s;
  per = (per + 255) / 256 * 256;
  return (int)(((long long)r + per - 1) / per);
}

constexpr int kF4MomentParts = 256;  // = lanes of the combine kernel

}  // namespace

#define MLP_API extern "C" __attribute__((visibility("default")))

// doubles of a moments buffer (256 partial rows of 14 sums)
MLP_API int mlp_first4_moments_doubles(void) { return kF4MomentParts * 14; }

// the 14 moments of x (b,4,r) -- 4 sums, 10 products -- as 256 partial rows of doubles
MLP_API int mlp_first4_moments(int b, int r, const float *x, double *moments, void *stream_) {
  if (b <= 0 || r <= 0) return 0;
  if (!moments) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(first4_moments_kernel, dim3(kF4MomentParts), dim3(256), 0, (hipStream_t)stream_, r,
                     (long long)b * r, x, moments);
  return pn2_launch_status();
}

// training-mode BatchNorm coefficients (and running-statistics update) of y = w x, w (64,4), from
// the moments of x over count = b*r columns; outputs as mlp_bn_finalize_pairs
MLP_API int mlp_first4_bn(const double *moments, double count, const float *w, const float *gamma,
                          const float *beta, float eps, float momentum, float *running_mean,
                          float *running_var, float *mean, float *invstd, float *scale,
                          float *shift, void *stream_) {
  if (!moments || count <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(first4_bn_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, moments, count, w,
                     gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  return pn2_launch_status();
}

// workspace of mlp_wgrad_first4: G partials + their sum (floats), then moment partials (doubles)
MLP_API size_t mlp_wgrad_first4_workspace_bytes(int b, int r) {
  if (b <= 0 || r <= 0) return 0;
  const size_t g = sizeof(float) * ((size_t)b * first4_slices(b, r) + 1) * kF4Rows * 4;
  return (g + 7) / 8 * 8 + sizeof(double) * kF4MomentParts * 14;
}

// dw (64,4) of the layer y = w x, x (b,4,r), behind BatchNorm + ReLU, from dz (b,64,r) = the
// gradient w.r.t. relu(bn(y)) and the layer's (scale, shift, mean, invstd, coef (64,3)); y itself
// is not read.  r % 4 == 0.  moments: those of x from mlp_first4_moments (the forward's), or NULL
// to compute them here.
MLP_API int mlp_wgrad_first4(int b, int r, const float *w, const float *x, const float *dz,
                             const float *scale, const float *shift, const float *mean,
                             const float *invstd, const float *coef, const double *moments,
                             float *dw, void *workspace, void *stream_) {
  if (b <= 0 || r <= 0) return 0;
  if (r % 4 != 0 || !workspace) return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  const int slices = first4_slices(b, r);
  const long long per = (((long long)r + slices - 1) / slices + 255) / 256 * 256;
  float *gpart = static_cast<float *>(workspace);
  float *gsum = gpart + (size_t)b * slices * kF4Rows * 4;
  const size_t gbytes = (sizeof(float) * ((size_t)b * slices + 1) * kF4Rows * 4 + 7) / 8 * 8;
  double *mpart = reinterpret_cast<double *>(static_cast<char *>(workspace) + gbytes);
  hipLaunchKernelGGL(first4_gated_kernel, dim3(slices, b), dim3(512), 0, stream, r, (int)per, w, x, dz,
                     scale, shift, gpart);
  if (moments == nullptr) {
    hipLaunchKernelGGL(first4_moments_kernel, dim3(kF4MomentParts), dim3(256), 0, stream, r,
                       (long long)b * r, x, mpart);
    moments = mpart;
  }
  int rc = mlp_reduce_partials(kF4Rows * 4, b * slices, gpart, gsum, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(first4_combine_kernel, dim3(1), dim3(256), 0, stream, gsum, moments, w, mean, invstd,
                     coef, dw);
  return pn2_launch_status();
}

// The same weight gradient when the gated sums G come from elsewhere: the one-pass backward of the
// layer ABOVE (mlp_gemm_backward_fused, qmode 4) forms dz -- the gradient w.r.t. this virtual layer's
// activated output -- block by block in its accumulators and leaves G as `parts` partials of (64,4)
// instead of writing dz (268 MB at SA1) for mlp_wgrad_first4 to read.  moments: mlp_first4_moments of
// x (required); workspace: 256 floats.
MLP_API int mlp_wgrad_first4_from_gated(int parts, const float *gpart, const float *w, const float *mean,
                                        const float *invstd, const float *coef, const double *moments,
                                        float *dw, float *workspace, void *stream_) {
  if (parts <= 0 || !gpart || !w || !mean || !invstd || !coef || !moments || !dw || !workspace)
    return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = mlp_reduce_partials(kF4Rows * 4, parts, gpart, workspace, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(first4_combine_kernel, dim3(1), dim3(256), 0, stream, workspace, moments, w, mean, invstd,
                     coef, dw);
  return pn2_launch_status();
}

