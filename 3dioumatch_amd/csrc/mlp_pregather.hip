// 3dioumatch_amd/csrc/mlp_pregather.hip -- the FIRST layer of a set-abstraction module's shared
// MLP, applied BEFORE the gather.
//
// Reference data flow (pointnet2_utils.py:335-358 QueryAndGroup, then pytorch_utils.py:14-39):
//   grouped (B, 3+C, m, ns) = [ (xyz[idx] - new_xyz) * s ; features[idx] ]      -- written,
//   y1 = W1 . grouped                                                             -- read again.
// A 1x1 convolution commutes with the gather:  with  src = [xyz*s ; features]  (B, 3+C, N)
//   y1[:, j, t] = (W1 . src)[:, idx[j, t]]  -  (W1[:, :3] . new_xyz*s)[:, j]
// so the GEMM runs over the N points of the cloud instead of the m*ns gathered columns (16x fewer
// at SA2) and the (3+C)-row grouped tensor is never formed -- neither in the forward nor, as
// its gradient, in the backward.  The two terms come out of ONE GEMM over the packed operand
//   src_ext (B, 3+C, N+m) = [ xyz*s | new_xyz*s ]     z_ext = W1 . src_ext   (B, M, N+m)
//                           [ feat  |    0      ]
// and so do their gradients: dW1 = dz_ext . src_ext^T, d src_ext = W1^T . dz_ext.
//
//   pregather_pack_kernel      builds src_ext (one launch instead of a transpose, a pad and two cats)
//   pregather_forward_kernel   y1 = z_ext[idx] - z_ext[N + j], and the (mean, M2) pair of every
//                              (cloud, channel) row for the layer's BatchNorm
//   pregather_backward_kernel  dy1 from (y1, dz1) on the fly (the BatchNorm + ReLU backward of
//                              mlp_operand.h), scatter-added through the inverse index into
//                              dz_ext[:, :N] and summed per group into dz_ext[:, N + j] (negated)
//
// Same results as the grouped form up to fp32 summation order (3+C terms in a different order,
// and W.(a - b) as W.a - W.b): relative 1e-6, inside the 1e-4 feature bar (tests/test_gpu_mlp.py).
#include "common.h"
#include "mlp_operand.h"

namespace {

constexpr int kPgThreads = 1024;
constexpr int kPgMaxPoints = 4096;   // source row staged in LDS
constexpr int kPgMaxGroups = 4096;   // centroid row staged in LDS
constexpr int kPgCB = 4;             // channels per workgroup of the forward kernel

// src_ext[b, r, p]: r < 3: s * xyz[b, p, r] (p < n) or s * new_xyz[b, p - n, r]; r >= 3:
// features[b, r - 3, p] (p < n) or 0
__global__ void __launch_bounds__(256)
pregather_pack_kernel(int n, int m, int c, float s, const float *__restrict__ xyz,
                      const float *__restrict__ new_xyz, const float *__restrict__ features,
                      float *__restrict__ out) {
  const int p = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  const int w = n + m;
  if (p >= w) return;
  float v;
  if (r < 3) v = s * (p < n ? xyz[((size_t)b * n + p) * 3 + r] : new_xyz[((size_t)b * m + (p - n)) * 3 + r]);
  else v = p < n ? features[((size_t)b * c + (r - 3)) * n + p] : 0.f;
  out[((size_t)b * (3 + c) + r) * w + p] = v;
}

// the reverse for the gradient: d features (b, c, n) = d src_ext[:, 3:, :n] (contiguous copy)
__global__ void __launch_bounds__(256)
pregather_unpack_kernel(int n, int m, int c, const float *__restrict__ dsrc, float *__restrict__ dfeat) {
  const int p = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (p >= n) return;
  dfeat[((size_t)b * c + r) * n + p] = dsrc[((size_t)b * (3 + c) + 3 + r) * (size_t)(n + m) + p];
}

__device__ __forceinline__ float block_sum(float v, float *scratch) {  // all threads get the sum
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
#pragma unroll
  for (int o = kWave / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
  __syncthreads();
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int q = 0; q < kPgThreads / kWave; ++q) t += scratch[q];
  return t;
}

// y[b, ch, j, t] = z[b, ch, idx[b, j, t]] - z[b, ch, n + j]; z rows have n + m columns.
// pairs (b, c, 2) or null: (mean, M2) of every (cloud, channel) row of y.
__global__ void __launch_bounds__(kPgThreads)
pregather_forward_kernel(int c, int n, int m, int ns, const float *__restrict__ z,
                         const int *__restrict__ idx, float *__restrict__ y,
                         float *__restrict__ pairs) {
  __shared__ __attribute__((aligned(16))) float zrow[kPgCB][kPgMaxPoints];
  __shared__ __attribute__((aligned(16))) float vrow[kPgCB][kPgMaxGroups];
  __shared__ float scratch[kPgThreads / kWave];
  const BlockId blk = xcd_block_id();  // the channel blocks of a cloud share its idx in one L2
  const int ch0 = blk.x * kPgCB, b = blk.y, tid = threadIdx.x;
  const int w = n + m, mns = m * ns;
#pragma unroll
  for (int q = 0; q < kPgCB; ++q) {
    const float *row = z + ((size_t)b * c + ch0 + q) * w;
    for (int t = tid; t < n; t += kPgThreads) zrow[q][t] = row[t];
    for (int t = tid; t < m; t += kPgThreads) vrow[q][t] = row[n + t];
  }
  __syncthreads();
  const int4 *i4 = reinterpret_cast<const int4 *>(idx + (size_t)b * mns);
  float a1[kPgCB], a2[kPgCB], sh[kPgCB];
#pragma unroll
  for (int q = 0; q < kPgCB; ++q) {
    a1[q] = 0.f; a2[q] = 0.f;
    sh[q] = zrow[q][idx[(size_t)b * mns]] - vrow[q][0];  // shifted sums: y[b, ch, 0, 0]
  }
  // m * ns <= 32768: at most eight 16-byte index loads per lane, all in flight before the first use
  const int n4 = mns / 4;
  int4 iv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e4 = tid + u * kPgThreads;
    iv[u] = e4 < n4 ? i4[e4] : make_int4(0, 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e4 = tid + u * kPgThreads;
    if (e4 < n4) {  // ns % 4 == 0: the four share a group
      const int4 ii = iv[u];
      const int j = (e4 * 4) / ns;
#pragma unroll
      for (int q = 0; q < kPgCB; ++q) {
        const float vv = vrow[q][j];
        float4 o;
        o.x = zrow[q][ii.x] - vv; o.y = zrow[q][ii.y] - vv;
        o.z = zrow[q][ii.z] - vv; o.w = zrow[q][ii.w] - vv;
        reinterpret_cast<float4 *>(y + ((size_t)b * c + ch0 + q) * mns)[e4] = o;
        const float d0 = o.x - sh[q], d1 = o.y - sh[q], d2 = o.z - sh[q], d3 = o.w - sh[q];
        a1[q] += (d0 + d1) + (d2 + d3);
        a2[q] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
  }
  if (pairs == nullptr) return;
#pragma unroll
  for (int q = 0; q < kPgCB; ++q) {
    const float s1 = block_sum(a1[q], scratch), s2 = block_sum(a2[q], scratch);
    if (tid == 0) {
      const float fn = (float)mns;
      float *out = pairs + ((size_t)b * c + ch0 + q) * 2;
      out[0] = sh[q] + s1 / fn;
      out[1] = s2 - s1 * s1 / fn;
    }
  }
}

// One workgroup per (channel, cloud).  row[e] = dy1[b, ch, e] from (y, dz) with the row's
// BatchNorm / ReLU backward constants; dzx[b, ch, p] (p < n) = sum of row[e] over idx[e] == p,
// through the inverse index of pn2_group_inverse_build (pn2_ball_group.hip: entry s of the
// sorted order at [(s % CHUNK) * 1024 + s / CHUNK], point << 16 | position);
// dzx[b, ch, n + j] = -sum_t row[j * ns + t].
template <int CHUNK>
__global__ void __launch_bounds__(kPgThreads)
pregather_backward_kernel(int c, int n, int m, int ns, OperandB op,
                          const unsigned *__restrict__ inv, float *__restrict__ dzx) {
  __shared__ __attribute__((aligned(16))) float row[CHUNK * 1024];
  __shared__ float acc[kPgMaxPoints];
  const BlockId blk = xcd_block_id();
  const int ch = blk.x, b = blk.y, tid = threadIdx.x;
  const int mns = m * ns, w = n + m;
  const unsigned *ent = inv + (size_t)b * CHUNK * 1024 + tid;
  unsigned e[CHUNK];
#pragma clang loop unroll(full)
  for (int j = 0; j < CHUNK; ++j) e[j] = ent[j * 1024];
  const RowCoef rc = load_row_coef<OP_DY>(op, ch, true);
  const size_t off = ((size_t)b * c + ch) * mns;
  {
    const float4 *y4 = reinterpret_cast<const float4 *>(op.x + off);
    const float4 *d4 = reinterpret_cast<const float4 *>(op.dz + off);
    float4 *r4 = reinterpret_cast<float4 *>(row);
    for (int t0 = tid; t0 < mns / 4; t0 += 4 * kPgThreads) {  // eight 16-byte loads in flight
      float4 yv[4], dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * kPgThreads;
        if (t < mns / 4) { yv[u] = y4[t]; dv[u] = d4[t]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * kPgThreads;
        if (t < mns / 4) {
          float4 o;
          o.x = transform<OP_DY>(yv[u].x, dv[u].x, rc); o.y = transform<OP_DY>(yv[u].y, dv[u].y, rc);
          o.z = transform<OP_DY>(yv[u].z, dv[u].z, rc); o.w = transform<OP_DY>(yv[u].w, dv[u].w, rc);
          r4[t] = o;
        }
      }
    }
  }
  for (int t = tid; t < n; t += kPgThreads) acc[t] = 0.f;
  __syncthreads();
  float *dst = dzx + ((size_t)b * c + ch) * w;
  for (int j = tid; j < m; j += kPgThreads) {  // group sums; the rotation keeps the lanes of a
    float s = 0.f;                             // wave on different banks
    for (int t = 0; t < ns; ++t) s += row[j * ns + ((t + j) & (ns - 1))];
    dst[n + j] = -s;
  }
  float v[CHUNK];
#pragma clang loop unroll(full)
  for (int j = 0; j < CHUNK; ++j) v[j] = row[e[j] & (CHUNK * 1024 - 1)];
  unsigned cur = e[0] >> 16;
  float sum = v[0];
  bool shared_run = true;  // the lane's first run may have begun in the lane before
#pragma clang loop unroll(full)
  for (int j = 1; j < CHUNK; ++j) {
    const unsigned key = e[j] >> 16;
    if (key != cur) {
      if (shared_run) atomicAdd(&acc[cur], sum); else acc[cur] = sum;
      shared_run = false;
      sum = 0.f;
      cur = key;
    }
    sum = __fadd_rn(sum, v[j]);
  }
  if (cur != (0xFFFFFFFFu >> 16)) atomicAdd(&acc[cur], sum);  // may continue in the next lane
  __syncthreads();
  for (int t = tid; t < n; t += kPgThreads) dst[t] = acc[t];
}

bool pregather_shape_ok(int b, int c, int n, int m, int ns) {
  const long long mns = (long long)m * ns;
  return b > 0 && c > 0 && c % kPgCB == 0 && c <= 65535 && n > 0 && n <= kPgMaxPoints && m > 0 &&
         m <= kPgMaxGroups && ns >= 4 && (ns & (ns - 1)) == 0 && mns <= 32768;
}

}  // namespace

#define MLP_API extern "C" __attribute__((visibility("default")))

MLP_API int mlp_pregather_supported(int b, int c, int n, int m, int ns) {
  return pregather_shape_ok(b, c, n, m, ns) ? 1 : 0;
}

MLP_API int mlp_pregather_pack(int b, int n, int m, int c, float s, const float *xyz,
                               const float *new_xyz, const float *features, float *src_ext,
                               void *stream) {
  if (b <= 0 || n <= 0 || m <= 0 || c < 0) return (int)hipErrorInvalidValue;
  if (c > 65532) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(pregather_pack_kernel, dim3(pn2_ceil_div(n + m, 256), 3 + c, b), dim3(256), 0,
                     (hipStream_t)stream, n, m, c, s, xyz, new_xyz, features, src_ext);
  return pn2_launch_status();
}

MLP_API int mlp_pregather_unpack_grad(int b, int n, int m, int c, const float *dsrc_ext,
                                      float *dfeatures, void *stream) {
  if (b <= 0 || n <= 0 || m <= 0 || c <= 0 || c > 65535) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(pregather_unpack_kernel, dim3(pn2_ceil_div(n, 256), c, b), dim3(256), 0,
                     (hipStream_t)stream, n, m, c, dsrc_ext, dfeatures);
  return pn2_launch_status();
}

MLP_API int mlp_pregather_forward(int b, int c, int n, int m, int ns, const float *z_ext,
                                  const int *idx, float *y, float *pairs, void *stream) {
  if (!pregather_shape_ok(b, c, n, m, ns)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(pregather_forward_kernel, dim3(c / kPgCB, b), dim3(kPgThreads), 0,
                     (hipStream_t)stream, c, n, m, ns, z_ext, idx, y, pairs);
  return pn2_launch_status();
}

MLP_API int mlp_pregather_backward(int b, int c, int n, int m, int ns, const float *y,
                                   const float *dz, const float *scale, const float *shift,
                                   const float *mean, const float *invstd, const float *coef,
                                   const unsigned *inverse, float *dz_ext, void *stream_) {
  if (!pregather_shape_ok(b, c, n, m, ns)) return (int)hipErrorInvalidValue;
  OperandB op = {y, dz, scale, shift, mean, invstd, coef, nullptr, 0, 0, nullptr};
  const dim3 grid(c, b);
  hipStream_t stream = (hipStream_t)stream_;
  int chunk = 4;
  while (chunk * 1024 < m * ns) chunk *= 2;  // inverse_chunk() of pn2_ball_group.hip
  switch (chunk) {
    case 4:
      hipLaunchKernelGGL(pregather_backward_kernel<4>, grid, dim3(kPgThreads), 0, stream, c, n, m, ns,
                         op, inverse, dz_ext);
      break;
    case 8:
      hipLaunchKernelGGL(pregather_backward_kernel<8>, grid, dim3(kPgThreads), 0, stream, c, n, m, ns,
                         op, inverse, dz_ext);
      break;
    case 16:
      hipLaunchKernelGGL(pregather_backward_kernel<16>, grid, dim3(kPgThreads), 0, stream, c, n, m, ns,
                         op, inverse, dz_ext);
      break;
    default:
      hipLaunchKernelGGL(pregather_backward_kernel<32>, grid, dim3(kPgThreads), 0, stream, c, n, m, ns,
                         op, inverse, dz_ext);
  }
  return pn2_launch_status();
}
