// 3dioumatch_amd/csrc/pn2_interp.hip -- three_nn / three_interpolate (+grad) for gfx950.
//
// Semantics: reference interpolate_gpu.cu:14-64 (K7), :77-106 (K8), :121-148 (K9, the
// intended backward that the reference never dispatches); SURVEY App. A.5-A.7.
// The reference uses one block per cloud.  Here:
//  * three_nn: (n/256, B) workgroups; the known cloud is staged through LDS as float4
//    (one ds_read_b128 broadcast per test) and every lane owns one query point.  The
//    3-slot insertion keeps the reference's strict '<' chain, so the earliest index wins ties.
//  * three_interpolate: lanes own consecutive query points j (coalesced idx/weight/out
//    traffic), and walk a group of channels so idx/weight are fetched once.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kNNTile = 1024;  // known points per LDS stage (16 KiB)

__global__ void __launch_bounds__(256)
three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                const float *__restrict__ known, float *__restrict__ dist2,
                int *__restrict__ idx) {
  __shared__ float4 tile[kNNTile];
  const BlockId blk = xcd_block_id();
  const int b = blk.y;
  const int j = blk.x * 256 + threadIdx.x;
  const bool live = j < n;
  const float *kn = known + (size_t)b * m * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (live) {
    const float *u = unknown + ((size_t)b * n + j) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  // the reference's accumulators are doubles initialised to 1e40; for fp32 candidates that
  // is indistinguishable from +inf, and (float)1e40 == +inf on output
  float best1 = __builtin_inff(), best2 = __builtin_inff(), best3 = __builtin_inff();
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int base = 0; base < m; base += kNNTile) {
    const int cnt = m - base < kNNTile ? m - base : kNNTile;
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      const float *p = kn + (size_t)(base + t) * 3;
      tile[t] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
#pragma unroll 4
    for (int t = 0; t < cnt; ++t) {
      const float4 p = tile[t];
      const float d = sqdist3(ux, uy, uz, p.x, p.y, p.z);
      if (d < best3) {
        const int k = base + t;
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else {
          best3 = d; besti3 = k;
        }
      }
    }
  }
  if (live) {
    float *od = dist2 + ((size_t)b * n + j) * 3;
    int *oi = idx + ((size_t)b * n + j) * 3;
    od[0] = best1; od[1] = best2; od[2] = best3;
    oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
  }
}

// The same for launches that do not fill the chip (the IoU branch's grid points: 9216 queries x 8
// clouds = 288 workgroups of the kernel above, ONE wave per SIMD, every candidate a dependent chain
// of an LDS round trip and ten vector instructions: 125 us for 75 M tests).  Here a workgroup owns 64
// queries and its four waves each scan a QUARTER of every tile for all of them -- four times the waves
// in flight -- then wave 0 folds the quarters' sorted triples in, quarter by quarter in index order,
// with the same strict '<' insertion: a later quarter holds larger indices only, so the earliest
// index still wins every tie and the result is the sequential scan's, bit for bit.
__global__ void __launch_bounds__(256)
three_nn_split_kernel(int n, int m, const float *__restrict__ unknown,
                      const float *__restrict__ known, float *__restrict__ dist2,
                      int *__restrict__ idx) {
  __shared__ float4 tile[kNNTile];
  __shared__ float pd[3][3][kWave];
  __shared__ int pi[3][3][kWave];
  const BlockId blk = xcd_block_id();
  const int b = blk.y;
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int j = blk.x * kWave + lane;
  const bool live = j < n;
  const float *kn = known + (size_t)b * m * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (live) {
    const float *u = unknown + ((size_t)b * n + j) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  float best1 = __builtin_inff(), best2 = __builtin_inff(), best3 = __builtin_inff();
  int besti1 = 0, besti2 = 0, besti3 = 0;
  // (branch-free: in a scan of 256 candidates some lane of the wave improves its triple at almost
  // every step, so the branchy form runs its slow path nearly always -- and diverged)
  auto insert = [&](float d, int k) {
    const bool c1 = d < best1, c2 = d < best2, c3 = d < best3;
    best3 = c2 ? best2 : (c3 ? d : best3);
    besti3 = c2 ? besti2 : (c3 ? k : besti3);
    best2 = c1 ? best1 : (c2 ? d : best2);
    besti2 = c1 ? besti1 : (c2 ? k : besti2);
    best1 = c1 ? d : best1;
    besti1 = c1 ? k : besti1;
  };
  for (int base = 0; base < m; base += kNNTile) {
    const int cnt = m - base < kNNTile ? m - base : kNNTile;
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      const float *p = kn + (size_t)(base + t) * 3;
      tile[t] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
    // this wave's quarter of the tile (whole tiles before it were folded into every wave's triple
    // in index order already: a wave's triple is over its quarters of ALL tiles so far -- the final
    // fold below therefore goes tile-major only when m <= kNNTile; larger m takes the plain kernel)
    const int q = (cnt + 3) / 4;
    const int t0 = w * q, t1 = t0 + q < cnt ? t0 + q : cnt;
#pragma unroll 4
    for (int t = t0; t < t1; ++t) {
      const float4 p = tile[t];
      insert(sqdist3(ux, uy, uz, p.x, p.y, p.z), base + t);
    }
  }
  if (w > 0) {
    pd[w - 1][0][lane] = best1; pd[w - 1][1][lane] = best2; pd[w - 1][2][lane] = best3;
    pi[w - 1][0][lane] = besti1; pi[w - 1][1][lane] = besti2; pi[w - 1][2][lane] = besti3;
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int qq = 0; qq < 3; ++qq)
#pragma unroll
      for (int s = 0; s < 3; ++s) insert(pd[qq][s][lane], pi[qq][s][lane]);
    if (live) {
      float *od = dist2 + ((size_t)b * n + j) * 3;
      int *oi = idx + ((size_t)b * n + j) * 3;
      od[0] = best1; od[1] = best2; od[2] = best3;
      oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
  }
}

// out[b,l,j] = p[i1]*w1 + p[i2]*w2 + p[i3]*w3, left to right (interpolate_gpu.cu:77-106).
// A lane owns JP consecutive query points (JP = 4 -> one 16-byte store per channel) and
// walks a group of channels, so 12 independent gathers are in flight per channel step.
template <int JP>
__global__ void __launch_bounds__(256)
three_interpolate_kernel(int c, int m, int n, const float *__restrict__ points,
                         const int *__restrict__ idx, const float *__restrict__ weight,
                         float *__restrict__ out, size_t out_bstride) {
  const BlockId blk = xcd_block_id();
  const int b = blk.z;
  const int j0 = (blk.x * 256 + threadIdx.x) * JP;
  if (j0 >= n) return;
  int ii[JP][3];
  float ww[JP][3];
#pragma unroll
  for (int t = 0; t < JP; ++t) {
    const int j = j0 + t < n ? j0 + t : n - 1;
    const int *ib = idx + ((size_t)b * n + j) * 3;
    const float *wb = weight + ((size_t)b * n + j) * 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) { ii[t][q] = ib[q]; ww[t][q] = wb[q]; }
  }
  for (int l = blk.y; l < c; l += gridDim.y) {
    const float *src = points + ((size_t)b * c + l) * m;
    float r[JP];
#pragma unroll
    for (int t = 0; t < JP; ++t)
      r[t] = __fadd_rn(__fadd_rn(__fmul_rn(src[ii[t][0]], ww[t][0]),
                                 __fmul_rn(src[ii[t][1]], ww[t][1])),
                       __fmul_rn(src[ii[t][2]], ww[t][2]));
    float *dst = out + (size_t)b * out_bstride + (size_t)l * n + j0;
    if (JP == 4) {
      *reinterpret_cast<float4 *>(dst) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
      dst[0] = r[0];
    }
  }
}

// The same with the CPW source rows of the workgroup staged in LDS (m*4 bytes each): the 3 x JP
// gathers per channel become LDS reads (the global form is bound by the texture-address unit:
// one 4-byte gather per address), and the kernel is left with its 16-byte output stores.  idx and
// weight of the lane's queries are read once per channel group (they stay in the XCD's L2).
template <int CPW>
__global__ void __launch_bounds__(256)
three_interpolate_lds_kernel(int c, int m, int n, const float *__restrict__ points,
                             const int *__restrict__ idx, const float *__restrict__ weight,
                             float *__restrict__ out, size_t out_bstride) {
  extern __shared__ __attribute__((aligned(16))) float rows[];
  const BlockId blk = xcd_block_id();
  const int b = blk.z, l0 = blk.y * CPW;
  const int nc = c - l0 < CPW ? c - l0 : CPW;
  const float *src = points + ((size_t)b * c + l0) * m;
  for (int t = threadIdx.x; t < nc * m; t += 256) rows[t] = src[t];
  __syncthreads();
  const int j0 = (blk.x * 256 + threadIdx.x) * 4;
  if (j0 >= n) return;
  const int4 *ib = reinterpret_cast<const int4 *>(idx + ((size_t)b * n + j0) * 3);
  const float4 *wb = reinterpret_cast<const float4 *>(weight + ((size_t)b * n + j0) * 3);
  const int4 i0 = ib[0], i1 = ib[1], i2 = ib[2];
  const float4 w0 = wb[0], w1 = wb[1], w2 = wb[2];
  const int ii[12] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w};
  const float ww[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
  for (int cc = 0; cc < CPW; ++cc) {
    if (cc < nc) {
      const float *row = rows + cc * m;
      float r[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        r[t] = __fadd_rn(__fadd_rn(__fmul_rn(row[ii[3 * t]], ww[3 * t]),
                                   __fmul_rn(row[ii[3 * t + 1]], ww[3 * t + 1])),
                         __fmul_rn(row[ii[3 * t + 2]], ww[3 * t + 2]));
      // (n >= 2048 on this path: tens of MB read next by a GEMM from HBM anyway -> streaming store)
      typedef float ti_f4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(ti_f4{r[0], r[1], r[2], r[3]},
                                  reinterpret_cast<ti_f4 *>(out + (size_t)b * out_bstride + (size_t)(l0 + cc) * n + j0));
    }
  }
}

// three_interpolate_lds_kernel + an affine term in three more inputs per query:
//   out[b][l][j] = sum_q w[b][j][q] * points[b][l][idx[b][j][q]] + aw[l] . ax[b][:, j]
// What it is for: a 1x1 convolution W over cat([ax (3 rows), interpolate(f)]) commutes with the
// interpolation -- W[:, 3:] . interpolate(f) = interpolate(W[:, 3:] . f) -- so the layer's GEMM runs
// over the m source points instead of the n >> m queries and this kernel writes the layer's output
// directly (the IoU branch's first layer, grid_conv_module.py:87-110: n = 64 grid points per box).
template <int CPW>
__global__ void __launch_bounds__(256)
three_interpolate_affine_lds_kernel(int c, int m, int n, const float *__restrict__ points,
                                    const int *__restrict__ idx, const float *__restrict__ weight,
                                    const float *__restrict__ aw, const float *__restrict__ ax,
                                    float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float rows[];
  const BlockId blk = xcd_block_id();
  const int b = blk.z, l0 = blk.y * CPW;
  const int nc = c - l0 < CPW ? c - l0 : CPW;
  const float *src = points + ((size_t)b * c + l0) * m;
  for (int t = threadIdx.x; t < nc * m; t += 256) rows[t] = src[t];
  __syncthreads();
  const int j0 = (blk.x * 256 + threadIdx.x) * 4;
  if (j0 >= n) return;
  const int4 *ib = reinterpret_cast<const int4 *>(idx + ((size_t)b * n + j0) * 3);
  const float4 *wb = reinterpret_cast<const float4 *>(weight + ((size_t)b * n + j0) * 3);
  const int4 i0 = ib[0], i1 = ib[1], i2 = ib[2];
  const float4 w0 = wb[0], w1 = wb[1], w2 = wb[2];
  const int ii[12] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w};
  const float ww[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
  const float *axb = ax + (size_t)b * 3 * n + j0;
  const float4 x0 = *reinterpret_cast<const float4 *>(axb), x1 = *reinterpret_cast<const float4 *>(axb + n),
               x2 = *reinterpret_cast<const float4 *>(axb + 2 * (size_t)n);
  const float xs[4][3] = {{x0.x, x1.x, x2.x}, {x0.y, x1.y, x2.y}, {x0.z, x1.z, x2.z}, {x0.w, x1.w, x2.w}};
#pragma unroll
  for (int cc = 0; cc < CPW; ++cc) {
    if (cc < nc) {
      const float *row = rows + cc * m;
      const float a0 = aw[(l0 + cc) * 3], a1 = aw[(l0 + cc) * 3 + 1], a2 = aw[(l0 + cc) * 3 + 2];
      float r[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float interp = __fadd_rn(__fadd_rn(__fmul_rn(row[ii[3 * t]], ww[3 * t]),
                                                 __fmul_rn(row[ii[3 * t + 1]], ww[3 * t + 1])),
                                       __fmul_rn(row[ii[3 * t + 2]], ww[3 * t + 2]));
        const float affine = __fadd_rn(__fadd_rn(__fmul_rn(a0, xs[t][0]), __fmul_rn(a1, xs[t][1])),
                                       __fmul_rn(a2, xs[t][2]));
        r[t] = __fadd_rn(affine, interp);
      }
      typedef float ti_f4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(ti_f4{r[0], r[1], r[2], r[3]},
                                  reinterpret_cast<ti_f4 *>(out + ((size_t)b * c + l0 + cc) * n + j0));
    }
  }
}

// Scatter-add with the CPW destination rows privatised in LDS (ds_add_f32), written back once:
// no global atomics and no pre-zeroing (the global-atomic kernel below remains for large m).
template <int CPW>
__global__ void __launch_bounds__(256)
three_interpolate_grad_lds_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                  size_t g_bstride, const int *__restrict__ idx,
                                  const float *__restrict__ weight,
                                  float *__restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float rows[];
  const BlockId blk = xcd_block_id();
  const int b = blk.y, l0 = blk.x * CPW;
  const int nc = c - l0 < CPW ? c - l0 : CPW;
  for (int t = threadIdx.x; t < nc * m; t += 256) rows[t] = 0.f;
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += 256) {
    const int *ib = idx + ((size_t)b * n + j) * 3;
    const float *wb = weight + ((size_t)b * n + j) * 3;
    const int i1 = ib[0], i2 = ib[1], i3 = ib[2];
    const float w1 = wb[0], w2 = wb[1], w3 = wb[2];
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
      if (cc < nc) {
        const float g = grad_out[(size_t)b * g_bstride + (size_t)(l0 + cc) * n + j];
        float *row = rows + cc * m;
        atomicAdd(row + i1, __fmul_rn(g, w1));
        atomicAdd(row + i2, __fmul_rn(g, w2));
        atomicAdd(row + i3, __fmul_rn(g, w3));
      }
    }
  }
  __syncthreads();
  float *dst = grad_points + ((size_t)b * c + l0) * m;
  for (int t = threadIdx.x; t < nc * m; t += 256) dst[t] = rows[t];
}

// grad_points[b,l,i_t] += grad_out[b,l,j] * w_t   (interpolate_gpu.cu:121-148)
__global__ void __launch_bounds__(256)
three_interpolate_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                              size_t g_bstride, const int *__restrict__ idx,
                              const float *__restrict__ weight, float *__restrict__ grad_points) {
  const BlockId blk = xcd_block_id();
  const int b = blk.z;
  const int j = blk.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int *ib = idx + ((size_t)b * n + j) * 3;
  const float *wb = weight + ((size_t)b * n + j) * 3;
  const int i1 = ib[0], i2 = ib[1], i3 = ib[2];
  const float w1 = wb[0], w2 = wb[1], w3 = wb[2];
  for (int l = blk.y; l < c; l += gridDim.y) {
    const float g = grad_out[(size_t)b * g_bstride + (size_t)l * n + j];
    float *dst = grad_points + ((size_t)b * c + l) * m;
    atomicAdd(dst + i1, __fmul_rn(g, w1));
    atomicAdd(dst + i2, __fmul_rn(g, w2));
    atomicAdd(dst + i3, __fmul_rn(g, w3));
  }
}

int interp_channel_groups(int c) {
  int g = (c + 7) / 8;
  if (g < 1) g = 1;
  if (g > 65535) g = 65535;
  return g;
}

}  // namespace

PN2_API int pn2_three_nn(int b, int n, int m, const float *unknown, const float *known,
                         float *dist2, int *idx, void *stream_) {
  if (b <= 0 || n <= 0) return 0;
  // a launch of the plain kernel that leaves most SIMDs with one wave or none: the split form
  // (one tile of candidates only: its fold assumes the quarters are in index order)
  static const bool split_off = getenv("PN2_THREE_NN_SPLIT") && atoi(getenv("PN2_THREE_NN_SPLIT")) == 0;
  // (measured: 8 x 9216 queries x 1024 candidates 107 -> 51 us, 8 x 2048 x 1024 104 -> 18; at
  // 8 x 32768 -- 1024 workgroups of the plain kernel, four per CU -- the branchy plain scan wins, 125
  // against 133 us: the threshold is two workgroups per CU)
  if (!split_off && m > 0 && m <= kNNTile && (long long)b * pn2_ceil_div(n, 256) <= 512) {
    dim3 grid(pn2_ceil_div(n, kWave), b);
    hipLaunchKernelGGL(three_nn_split_kernel, grid, dim3(256), 0, (hipStream_t)stream_, n, m, unknown,
                       known, dist2, idx);
    return pn2_launch_status();
  }
  dim3 grid(pn2_ceil_div(n, 256), b);
  hipLaunchKernelGGL(three_nn_kernel, grid, dim3(256), 0, (hipStream_t)stream_, n, m, unknown,
                     known, dist2, idx);
  return pn2_launch_status();
}

// out may be a channel slice of a wider (b, C_total, n) tensor: out_bstride = C_total * n floats
static int interpolate_run(int b, int c, int m, int n, const float *points, const int *idx,
                           const float *weight, float *out, size_t out_bstride,
                           hipStream_t stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  const bool vec = n % 4 == 0 && out_bstride % 4 == 0 && (reinterpret_cast<size_t>(out) & 15) == 0;
  if (vec && n >= 2048 && m > 0 && m <= 2048) {  // 8 source rows fit 64 KB of LDS
    constexpr int CPW = 8;
    dim3 grid(pn2_ceil_div(n, 1024), pn2_ceil_div(c, CPW), b);
    hipLaunchKernelGGL(three_interpolate_lds_kernel<CPW>, grid, dim3(256),
                       sizeof(float) * (size_t)CPW * m, stream, c, m, n, points, idx, weight, out,
                       out_bstride);
    return pn2_launch_status();
  }
  if (vec) {
    dim3 grid(pn2_ceil_div(n, 1024), interp_channel_groups(c), b);
    hipLaunchKernelGGL(three_interpolate_kernel<4>, grid, dim3(256), 0, stream, c, m, n, points,
                       idx, weight, out, out_bstride);
  } else {
    dim3 grid(pn2_ceil_div(n, 256), interp_channel_groups(c), b);
    hipLaunchKernelGGL(three_interpolate_kernel<1>, grid, dim3(256), 0, stream, c, m, n, points,
                       idx, weight, out, out_bstride);
  }
  return pn2_launch_status();
}

static int interpolate_grad_run(int b, int c, int n, int m, const float *grad_out, size_t g_bstride,
                                const int *idx, const float *weight, float *grad_points,
                                hipStream_t stream) {
  if (b <= 0 || c <= 0 || m <= 0) return 0;
  if (n > 0 && m <= 1024) {  // the destination rows of a workgroup fit its LDS
    // 4 channels per workgroup: the FP layers' 256 channels x 8 clouds are 512 workgroups (16
    // channels each left half the CUs idle with one 4-wave workgroup on the others: 50 us)
    constexpr int CPW = 4;
    hipLaunchKernelGGL(three_interpolate_grad_lds_kernel<CPW>, dim3(pn2_ceil_div(c, CPW), b),
                       dim3(256), sizeof(float) * (size_t)CPW * m, stream, c, n, m, grad_out,
                       g_bstride, idx, weight, grad_points);
    return pn2_launch_status();
  }
  const int e = pn2_zero_async(grad_points, sizeof(float) * (size_t)b * c * m, stream);
  if (e != 0) return e;
  if (n <= 0) return 0;
  dim3 grid(pn2_ceil_div(n, 256), interp_channel_groups(c), b);
  hipLaunchKernelGGL(three_interpolate_grad_kernel, grid, dim3(256), 0, stream, c, n, m, grad_out,
                     g_bstride, idx, weight, grad_points);
  return pn2_launch_status();
}

PN2_API int pn2_three_interpolate_affine_supported(int c, int m, int n) {
  return c > 0 && m > 0 && m <= 2048 && n >= 4 && n % 4 == 0;
}

PN2_API int pn2_three_interpolate_affine(int b, int c, int m, int n, const float *points, const int *idx,
                                         const float *weight, const float *affine_w,
                                         const float *affine_x, float *out, void *stream_) {
  if (b <= 0) return 0;
  if (!pn2_three_interpolate_affine_supported(c, m, n) || !affine_w || !affine_x ||
      (reinterpret_cast<size_t>(out) & 15) || (reinterpret_cast<size_t>(affine_x) & 15) ||
      (reinterpret_cast<size_t>(idx) & 15) || (reinterpret_cast<size_t>(weight) & 15))
    return (int)hipErrorInvalidValue;
  constexpr int CPW = 8;
  dim3 grid(pn2_ceil_div(n, 1024), pn2_ceil_div(c, CPW), b);
  hipLaunchKernelGGL(three_interpolate_affine_lds_kernel<CPW>, grid, dim3(256),
                     sizeof(float) * (size_t)CPW * m, (hipStream_t)stream_, c, m, n, points, idx, weight,
                     affine_w, affine_x, out);
  return pn2_launch_status();
}

PN2_API int pn2_three_interpolate(int b, int c, int m, int n, const float *points,
                                  const int *idx, const float *weight, float *out,
                                  void *stream_) {
  return interpolate_run(b, c, m, n, points, idx, weight, out, (size_t)c * n, (hipStream_t)stream_);
}

PN2_API int pn2_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                       const int *idx, const float *weight, float *grad_points,
                                       void *stream_) {
  return interpolate_grad_run(b, c, n, m, grad_out, (size_t)c * n, idx, weight, grad_points,
                              (hipStream_t)stream_);
}

// The same operators on channel SLICES of wider tensors: `out` / `grad_out` point at the first
// channel of the slice inside a (b, c_total, n) tensor.  The feature-propagation layer
// concatenates the interpolated features with the skip features (pointnet2_modules.py:404-410);
// writing the interpolation straight into the concatenated buffer, and reading its gradient
// straight out of the concatenated gradient, removes a full copy of both.
PN2_API int pn2_three_interpolate_into(int b, int c, int m, int n, const float *points,
                                       const int *idx, const float *weight, float *out,
                                       int c_total, void *stream_) {
  if (c_total < c) return (int)hipErrorInvalidValue;
  return interpolate_run(b, c, m, n, points, idx, weight, out, (size_t)c_total * n,
                         (hipStream_t)stream_);
}

PN2_API int pn2_three_interpolate_grad_from(int b, int c, int n, int m, const float *grad_out,
                                            int c_total, const int *idx, const float *weight,
                                            float *grad_points, void *stream_) {
  if (c_total < c) return (int)hipErrorInvalidValue;
  return interpolate_grad_run(b, c, n, m, grad_out, (size_t)c_total * n, idx, weight, grad_points,
                              (hipStream_t)stream_);
}
