// 3dioumatch_amd/csrc/mlp_bn.hip -- fused BatchNorm(+ReLU)(+max-pool over nsample) for the
// grouped shared MLP of the set-abstraction layers, forward and backward, gfx950.
//
// What it replaces: in the reference a shared-MLP layer is nn.Conv2d(1x1) -> nn.BatchNorm2d ->
// nn.ReLU (pointnet2/pytorch_utils.py:14-39,70-124) followed, after the last layer, by
// F.max_pool2d over the nsample axis (pointnet2/pointnet2_modules.py:256-262).  On GB-scale
// activations (B,C,npoint,nsample) that is, per layer, two reads + one write for the batch
// norm, one read + one write for the ReLU and one more read for the pooling, and about eight
// passes in the backward.  Here (training mode):
//   forward : stats (1 read) -> finalize (tiny) -> apply BN+ReLU (1 read, 1 write), or for the
//             last layer apply BN+ReLU+max over nsample (1 read, tiny write + arg-max);
//   backward: sums (read y, dz) -> finalize (tiny) -> dy (read y, dz; 1 write); the ReLU mask
//             and the normalised activation are RECOMPUTED from y, so neither z nor a mask
//             is stored; for the pooled layer the sums come from the (B,C,npoint) tensors alone.
// Statistics: per-slice shifted sums -> (n, mean, M2) -> Chan combination in double, i.e.
// Welford-quality means/variances independent of the slice count.
//
// Layout: y, z, dz, dy are (B, C, R) contiguous with R = npoint*nsample (or npoint for the
// 1-D case); per-channel vectors are length C.
#include "common.h"

namespace {

constexpr int kBnThreads = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = kWave / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// sum of two values over the workgroup; result valid in thread 0
__device__ __forceinline__ void block_sum2(float &a, float &b, float *scratch) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x / kWave;
  if (lane_id() == 0) { scratch[w * 2] = a; scratch[w * 2 + 1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = 0.f; b = 0.f;
    for (int q = 0; q < kBnThreads / kWave; ++q) { a += scratch[q * 2]; b += scratch[q * 2 + 1]; }
  }
}

// ---- "the last workgroup finalizes" -------------------------------------------------------------
// Every reduction here is partial sums per workgroup + a small finalize.  As two launches the
// finalize is a 4-5 us kernel of its own, ~90 of them per train step.  Instead every partial
// workgroup takes a ticket for its channel (an agent-scope release fence, then an atomic add) and
// the one that draws the last ticket reads all partials of the channel back -- in the same fixed
// order as the separate kernel, so the result does not depend on who is last -- and finalizes.
// The counters return to zero (the last workgroup resets its own), so launches that are ordered
// one after the other need no memset and may share an array.  The array is the CALLER's
// (include/mlp_hip.h, `tickets`): the library keeps no counters of its own and no table keyed by
// stream -- two launches that may overlap (two streams, two graphs replayed side by side) are
// given two arrays by whoever owns the modules they belong to.

// Partials travel between workgroups (possibly on different XCDs, whose L2s are not coherent with
// one another) as agent-scope RELAXED atomics: write-through stores (sc1) and cache-bypassing
// loads.  An agent-scope fence instead would write back and invalidate the whole L2 of the XCD
// per workgroup (buffer_wbl2 / buffer_inv): measured, that made the train step 1.8 ms SLOWER.
template <typename T>
__device__ __forceinline__ void coherent_store(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T coherent_load(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// true in every thread of the workgroup that arrived last among `expected` for counter `t`
// (the partial of this workgroup was written by its thread 0 with coherent_store)
__device__ __forceinline__ bool last_arrival(int *t, int expected) {
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0);  // the write-through stores of the partial have been acknowledged
    asm volatile("" ::: "memory");
    const int seen = __hip_atomic_fetch_add(t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = seen == expected - 1;
    if (is_last) coherent_store(t, 0);
  }
  __syncthreads();
  return is_last != 0;
}

// ---- forward statistics: one (n, mean, M2) triple per (channel, batch, slice) ---------------
// ---- finalize: Chan combination, running statistics, affine coefficients ------------------
// One wave per channel: lanes combine every 64th partial, then a butterfly merges the lanes.
struct Moments { double n, mean, m2; };

__device__ __forceinline__ Moments chan_merge(const Moments &a, const Moments &b) {
  if (b.n <= 0.0) return a;
  if (a.n <= 0.0) return b;
  Moments o;
  const double delta = b.mean - a.mean;
  o.n = a.n + b.n;
  o.mean = a.mean + delta * b.n / o.n;
  o.m2 = a.m2 + b.m2 + delta * delta * a.n * b.n / o.n;
  return o;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask, kWave);
  hi = __shfl_xor(hi, mask, kWave);
  return __hiloint2double(hi, lo);
}

struct FwdFinalize {
  int *tickets;  // one counter per channel, zero between launches (the caller's array)
  const float *gamma, *beta;
  float eps, momentum;
  float *running_mean, *running_var, *mean_out, *invstd_out, *scale_out, *shift_out;
};

__device__ __forceinline__ void fwd_write_channel(int ch, double n, double mean, double m2, const FwdFinalize &f);

// one wave: Chan-merge the `parts` triples of channel ch, write its statistics and coefficients
__device__ __forceinline__ void fwd_finalize_channel(int ch, int parts, const float *__restrict__ partial,
                                                     const FwdFinalize &f) {
  const int lane = lane_id();
  Moments acc = {0.0, 0.0, 0.0};
  const float *p = partial + (size_t)ch * parts * 3;
  for (int q = lane; q < parts; q += kWave) {
    const Moments part = {(double)coherent_load(p + q * 3), (double)coherent_load(p + q * 3 + 1),
                          (double)coherent_load(p + q * 3 + 2)};
    acc = chan_merge(acc, part);
  }
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    Moments other = {shfl_xor_f64(acc.n, off), shfl_xor_f64(acc.mean, off),
                     shfl_xor_f64(acc.m2, off)};
    // merge in a lane-independent order so that every lane ends with the same bits
    acc = (lane & off) ? chan_merge(other, acc) : chan_merge(acc, other);
  }
  if (lane != 0) return;
  fwd_write_channel(ch, acc.n, acc.mean, acc.m2, f);
}

// one lane: the channel's statistics and coefficients from (n, mean, M2)
__device__ __forceinline__ void fwd_write_channel(int ch, double n, double mean, double m2, const FwdFinalize &f) {
  const double var = n > 0.0 ? m2 / n : 0.0;  // biased, used for normalisation
  const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
  const float fmean = (float)mean;
  f.mean_out[ch] = fmean;
  f.invstd_out[ch] = invstd;
  const float sc = f.gamma[ch] * invstd;
  f.scale_out[ch] = sc;
  f.shift_out[ch] = f.beta[ch] - fmean * sc;
  if (f.running_mean != nullptr) {  // nn.BatchNorm: unbiased variance in the running estimate
    const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
    f.running_mean[ch] = (1.f - f.momentum) * f.running_mean[ch] + f.momentum * fmean;
    f.running_var[ch] = (1.f - f.momentum) * f.running_var[ch] + f.momentum * (float)unbiased;
  }
}

// ---- the small layers (FP modules, heads: at most 16384 values per channel): ONE workgroup per
// channel reads all of its values -- b rows of r floats, at most four float4 per lane of 1024 --, reduces
// them in double through LDS and finalizes.  No partials, no tickets, no second look at memory: the
// ticket form is three dependent trips to the memory side (~9-15 us per launch), this one load + one
// reduction (~5 us).  The backward form keeps y and dz in registers between the sums and dy.
constexpr int kChThreads = 1024, kChMax = 4 * 4 * kChThreads;  // values per channel
__device__ __forceinline__ void channel_sum2(double &a, double &b, double (*red)[2]) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) { a += shfl_xor_f64(a, off); b += shfl_xor_f64(b, off); }
  const int w = threadIdx.x / kWave;
  if (lane_id() == 0) { red[w][0] = a; red[w][1] = b; }
  __syncthreads();
  a = 0.0; b = 0.0;
#pragma unroll
  for (int q = 0; q < kChThreads / kWave; ++q) { a += red[q][0]; b += red[q][1]; }  // every thread: the same bits
}

__global__ void __launch_bounds__(kChThreads)
bn_channel_stats_kernel(int bn, int c, int r, const float *__restrict__ y, FwdFinalize fin) {
  __shared__ double red[kChThreads / kWave][2];
  const int ch = blockIdx.x, nv = r >> 2, total = bn * nv;
  const float shift = y[(size_t)ch * r];  // shifted sums: no cancellation in the variance
  float4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = threadIdx.x + u * kChThreads;
    v[u] = make_float4(shift, shift, shift, shift);
    if (i < total) {
      const int b = i / nv, j = i - b * nv;
      v[u] = reinterpret_cast<const float4 *>(y + ((size_t)b * c + ch) * r)[j];
    }
  }
  float a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float p0 = v[u].x - shift, p1 = v[u].y - shift, p2 = v[u].z - shift, p3 = v[u].w - shift;
    a1 += (p0 + p1) + (p2 + p3);
    a2 += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
  }
  double s1 = (double)a1, s2 = (double)a2;
  channel_sum2(s1, s2, red);
  if (threadIdx.x == 0) {
    const double n = (double)bn * (double)r;
    fwd_write_channel(ch, n, (double)shift + s1 / n, fmax(s2 - s1 * s1 / n, 0.0), fin);
  }
}

// the per-channel forms cover: whole float4 rows, at most kChMax values per channel, 16-byte aligned
// tensors (MLP_BN_CHANNEL_FORM=0: the ticket forms)
static bool channel_form(int b, int c, int r, const void *p0, const void *p1, const void *p2) {
  const char *env = getenv("MLP_BN_CHANNEL_FORM");  // (read on every call: the tests compare both forms)
  const bool off = env && atoi(env) == 0;
  if (off || r % 4 != 0 || (long long)b * r > kChMax || c < 32) return false;
  return ((reinterpret_cast<size_t>(p0) | reinterpret_cast<size_t>(p1) | reinterpret_cast<size_t>(p2)) & 15) == 0;
}

// ---- forward statistics (continued): the partial sums; the last workgroup of a channel also
// finalizes it (see last_arrival) ----------------------------------------------------------------
__global__ void __launch_bounds__(kBnThreads)
bn_partial_stats_kernel(int c, int r, int slices, const float *__restrict__ y,
                        float *__restrict__ partial, FwdFinalize fin) {
  __shared__ float scratch[2 * kBnThreads / kWave];
  const int s = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int per = (r + slices - 1) / slices;
  const int lo = s * per, hi = lo + per < r ? lo + per : r;
  const float *src = y + ((size_t)b * c + ch) * r;
  const int n = hi - lo;
  float a1 = 0.f, a2 = 0.f;
  const float shift = n > 0 ? src[lo] : 0.f;  // shifted sums: no cancellation in the variance
  if (((r | lo | hi) & 3) == 0) {
    // 16-byte loads, two in flight per lane (the row base b*c*r + ch*r is a multiple of 4 too)
    float b1 = 0.f, b2 = 0.f;
    const float4 *v = reinterpret_cast<const float4 *>(src + lo);
    const int nv = n >> 2;
    int i = threadIdx.x;
    for (; i + kBnThreads < nv; i += 2 * kBnThreads) {
      const float4 p = v[i], q = v[i + kBnThreads];
      const float p0 = p.x - shift, p1 = p.y - shift, p2 = p.z - shift, p3 = p.w - shift;
      const float q0 = q.x - shift, q1 = q.y - shift, q2 = q.z - shift, q3 = q.w - shift;
      a1 += (p0 + p1) + (p2 + p3);
      a2 += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
      b1 += (q0 + q1) + (q2 + q3);
      b2 += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
    }
    if (i < nv) {
      const float4 p = v[i];
      const float p0 = p.x - shift, p1 = p.y - shift, p2 = p.z - shift, p3 = p.w - shift;
      a1 += (p0 + p1) + (p2 + p3);
      a2 += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
    }
    a1 += b1;
    a2 += b2;
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += kBnThreads) {
      const float d = src[i] - shift;
      a1 += d;
      a2 += d * d;
    }
  }
  block_sum2(a1, a2, scratch);
  if (threadIdx.x == 0) {
    float *out = partial + ((size_t)ch * gridDim.z * slices + (size_t)b * slices + s) * 3;
    const float fn = (float)n;
    coherent_store(out, fn);
    coherent_store(out + 1, n > 0 ? shift + a1 / fn : 0.f);
    coherent_store(out + 2, n > 0 ? a2 - a1 * a1 / fn : 0.f);
  }
  if (last_arrival(&fin.tickets[ch], (int)gridDim.z * slices) && threadIdx.x < kWave)
    fwd_finalize_channel(ch, (int)gridDim.z * slices, partial, fin);
}

// The same from equal-count (mean, M2) pairs laid out [part][channel][2] (the epilogue of the
// forward GEMM, mlp_gemm.hip): mean = average of the part means, M2 = sum of the part M2 plus
// n_part * sum (mean_i - mean)^2, accumulated in double.  Stage 1: workgroup (channel block of
// 64, slice of the parts) -- lane = channel (512-byte contiguous reads per part), 16 waves split
// the slice, eight loads in flight per lane -- writes three double sums per channel and slice;
// stage 2 combines the slices.  (A single workgroup per channel block walked thousands of parts
// with one dependent load per step: 0.3 ms for SA1.)
constexpr int kPairSlices = 32;

// channel ch from its three sums over all parts (one lane per channel)
__device__ __forceinline__ void pairs_finalize_sums(int ch, int parts, int n_part, double s1, double s2,
                                                    double sm2, const float *__restrict__ pairs,
                                                    const FwdFinalize &f) {
  const double ref = (double)pairs[(size_t)ch * 2];
  const double P = (double)parts, n = P * (double)n_part;
  const double mean = ref + s1 / P;
  double m2 = sm2 + (double)n_part * (s2 - s1 * s1 / P);
  if (m2 < 0.0) m2 = 0.0;
  const double var = m2 / n;  // biased, used for normalisation
  const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
  const float fmean = (float)mean;
  f.mean_out[ch] = fmean;
  f.invstd_out[ch] = invstd;
  const float sc = f.gamma[ch] * invstd;
  f.scale_out[ch] = sc;
  f.shift_out[ch] = f.beta[ch] - fmean * sc;
  if (f.running_mean != nullptr) {  // nn.BatchNorm: unbiased variance in the running estimate
    const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
    f.running_mean[ch] = (1.f - f.momentum) * f.running_mean[ch] + f.momentum * fmean;
    f.running_var[ch] = (1.f - f.momentum) * f.running_var[ch] + f.momentum * (float)unbiased;
  }
}

// channel ch from the kPairSlices slice sums (one lane per channel)
__device__ __forceinline__ void pairs_finalize_channel(int ch, int c, int parts, int n_part,
                                                       const float *__restrict__ pairs,
                                                       const double *__restrict__ sums,
                                                       const FwdFinalize &f) {
  double s1 = 0.0, s2 = 0.0, sm2 = 0.0;
  for (int q = 0; q < kPairSlices; ++q) {
    const double *o = sums + ((size_t)q * c + ch) * 3;
    s1 += o[0]; s2 += o[1]; sm2 += o[2];
  }
  pairs_finalize_sums(ch, parts, n_part, s1, s2, sm2, pairs, f);
}

// (the ticket form was tried here too: the last slice's serial walk over 32 x 3 doubles per channel
//  cost more than the second launch it saved, 193 against 170 us per step -- two kernels stay)
__global__ void __launch_bounds__(256)
bn_pairs_stage2_kernel(int c, int parts, int n_part, const float *__restrict__ pairs,
                       const double *__restrict__ sums, FwdFinalize fin) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch < c) pairs_finalize_channel(ch, c, parts, n_part, pairs, sums, fin);
}

__global__ void __launch_bounds__(1024)
bn_pairs_stage1_kernel(int c, int parts, const float *__restrict__ pairs,
                       double *__restrict__ sums) {
  __shared__ double red[16][3][kWave];
  const int lane = lane_id(), w = threadIdx.x / kWave;
  const int ch = blockIdx.x * kWave + lane;
  const int per = (parts + kPairSlices - 1) / kPairSlices;
  const int p0 = blockIdx.y * per, p1 = p0 + per < parts ? p0 + per : parts;
  double s1 = 0.0, s2 = 0.0, sm2 = 0.0;
  if (ch < c) {
    const double ref = (double)pairs[(size_t)ch * 2];  // part 0's mean: no cancellation in s2
    int p = p0 + w;
    for (; p + 7 * 16 < p1; p += 8 * 16) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = *reinterpret_cast<const float2 *>(pairs + ((size_t)(p + u * 16) * c + ch) * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double d = (double)v[u].x - ref;
        s1 += d; s2 += d * d; sm2 += (double)v[u].y;
      }
    }
    for (; p < p1; p += 16) {
      const float2 v = *reinterpret_cast<const float2 *>(pairs + ((size_t)p * c + ch) * 2);
      const double d = (double)v.x - ref;
      s1 += d; s2 += d * d; sm2 += (double)v.y;
    }
  }
  red[w][0][lane] = s1; red[w][1][lane] = s2; red[w][2][lane] = sm2;
  __syncthreads();
  if (w == 0 && ch < c) {
    for (int q = 1; q < 16; ++q) { s1 += red[q][0][lane]; s2 += red[q][1][lane]; sm2 += red[q][2][lane]; }
    double *o = sums + ((size_t)blockIdx.y * c + ch) * 3;
    o[0] = s1; o[1] = s2; o[2] = sm2;
  }
}

// Few parts (the small layers: SA3 / SA4, the vote aggregation): ONE launch -- a workgroup per 64
// channels walks all parts (its 16 waves every 16th, eight loads in flight: <= 4 rounds at 512 parts)
// and finishes the channels itself.  Two launches cost ~11 us per layer whatever the size.
constexpr int kPairsOneLaunch = 512;
__global__ void __launch_bounds__(1024)
bn_pairs_one_kernel(int c, int parts, int n_part, const float *__restrict__ pairs, FwdFinalize fin) {
  __shared__ double red[16][3][kWave];
  const int lane = lane_id(), w = threadIdx.x / kWave;
  const int ch = blockIdx.x * kWave + lane;
  double s1 = 0.0, s2 = 0.0, sm2 = 0.0;
  if (ch < c) {
    const double ref = (double)pairs[(size_t)ch * 2];
    int p = w;
    for (; p + 7 * 16 < parts; p += 8 * 16) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = *reinterpret_cast<const float2 *>(pairs + ((size_t)(p + u * 16) * c + ch) * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double d = (double)v[u].x - ref;
        s1 += d; s2 += d * d; sm2 += (double)v[u].y;
      }
    }
    for (; p < parts; p += 16) {
      const float2 v = *reinterpret_cast<const float2 *>(pairs + ((size_t)p * c + ch) * 2);
      const double d = (double)v.x - ref;
      s1 += d; s2 += d * d; sm2 += (double)v.y;
    }
  }
  red[w][0][lane] = s1; red[w][1][lane] = s2; red[w][2][lane] = sm2;
  __syncthreads();
  if (w == 0 && ch < c) {
    for (int q = 1; q < 16; ++q) { s1 += red[q][0][lane]; s2 += red[q][1][lane]; sm2 += red[q][2][lane]; }
    pairs_finalize_sums(ch, parts, n_part, s1, s2, sm2, pairs, fin);
  }
}

// eval mode: coefficients from the running statistics
__global__ void __launch_bounds__(256)
bn_eval_coeff_kernel(int c, const float *__restrict__ gamma, const float *__restrict__ beta,
                     float eps, const float *__restrict__ running_mean,
                     const float *__restrict__ running_var, float *__restrict__ mean_out,
                     float *__restrict__ invstd_out, float *__restrict__ scale_out,
                     float *__restrict__ shift_out) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  const float invstd = 1.0f / sqrtf(running_var[ch] + eps);
  mean_out[ch] = running_mean[ch];
  invstd_out[ch] = invstd;
  const float sc = gamma[ch] * invstd;
  scale_out[ch] = sc;
  shift_out[ch] = beta[ch] - running_mean[ch] * sc;
}

// ---- z = relu(y*scale + shift) ---------------------------------------------------------------
template <bool VEC>
__global__ void __launch_bounds__(256)
bn_relu_apply_kernel(int c, int r, const float *__restrict__ y, const float *__restrict__ scale,
                     const float *__restrict__ shift, float *__restrict__ z) {
  const int ch = blockIdx.y, b = blockIdx.z;
  const float sc = scale[ch], sh = shift[ch];
  const size_t base = ((size_t)b * c + ch) * r;
  if (VEC) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= r) return;
    const float4 v = *reinterpret_cast<const float4 *>(y + base + i);
    float4 o;
    o.x = fmaxf(v.x * sc + sh, 0.f); o.y = fmaxf(v.y * sc + sh, 0.f);
    o.z = fmaxf(v.z * sc + sh, 0.f); o.w = fmaxf(v.w * sc + sh, 0.f);
    *reinterpret_cast<float4 *>(z + base + i) = o;
  } else {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < r) z[base + i] = fmaxf(y[base + i] * sc + sh, 0.f);
  }
}

// ---- pooled = max_k relu(y[..., j, k]*scale + shift), plus arg-max and the winning y ---------
// LPR lanes share one row of NS = 4*LPR samples (16-byte loads, coalesced over the wave);
// the row maximum is formed with xor-shuffles inside the LPR-lane group.
template <int LPR>
__global__ void __launch_bounds__(256)
bn_relu_pool_kernel(int c, int m, const float *__restrict__ y, const float *__restrict__ scale,
                    const float *__restrict__ shift, float *__restrict__ pooled,
                    int *__restrict__ argmax, float *__restrict__ ymax) {
  constexpr int NS = LPR * 4;
  const int ch = blockIdx.y, b = blockIdx.z;
  const int row = (blockIdx.x * 256 + threadIdx.x) / LPR;
  const int sub = threadIdx.x % LPR;
  const float sc = scale[ch], sh = shift[ch];
  const bool live = row < m;
  const size_t base = (((size_t)b * c + ch) * m + (live ? row : 0)) * NS + sub * 4;
  const float4 v = *reinterpret_cast<const float4 *>(y + base);
  float best = fmaxf(v.x * sc + sh, 0.f), by = v.x;
  int bk = sub * 4;
  float t;
  t = fmaxf(v.y * sc + sh, 0.f); if (t > best) { best = t; by = v.y; bk = sub * 4 + 1; }
  t = fmaxf(v.z * sc + sh, 0.f); if (t > best) { best = t; by = v.z; bk = sub * 4 + 2; }
  t = fmaxf(v.w * sc + sh, 0.f); if (t > best) { best = t; by = v.w; bk = sub * 4 + 3; }
#pragma unroll
  for (int off = 1; off < LPR; off <<= 1) {
    const float ob = __shfl_xor(best, off, kWave);
    const float oy = __shfl_xor(by, off, kWave);
    const int ok = __shfl_xor(bk, off, kWave);
    if (ob > best || (ob == best && ok < bk)) { best = ob; by = oy; bk = ok; }  // first max wins
  }
  if (live && sub == 0) {
    const size_t o = ((size_t)b * c + ch) * m + row;
    pooled[o] = best;
    argmax[o] = bk;
    ymax[o] = by;
  }
}

// generic nsample (not a multiple of 4 or > 64): one lane per row
__global__ void __launch_bounds__(256)
bn_relu_pool_generic_kernel(int c, int m, int ns, const float *__restrict__ y,
                            const float *__restrict__ scale, const float *__restrict__ shift,
                            float *__restrict__ pooled, int *__restrict__ argmax,
                            float *__restrict__ ymax) {
  const int ch = blockIdx.y, b = blockIdx.z;
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= m) return;
  const float sc = scale[ch], sh = shift[ch];
  const float *src = y + (((size_t)b * c + ch) * m + row) * ns;
  float best = fmaxf(src[0] * sc + sh, 0.f), by = src[0];
  int bk = 0;
  for (int k = 1; k < ns; ++k) {
    const float t = fmaxf(src[k] * sc + sh, 0.f);
    if (t > best) { best = t; by = src[k]; bk = k; }
  }
  const size_t o = ((size_t)b * c + ch) * m + row;
  pooled[o] = best; argmax[o] = bk; ymax[o] = by;
}

// pooled = max over nsample of relu(y*sc + sh) from the winning raw value per group that the
// forward GEMM left behind (the transform is monotone in y; the GEMM picked the largest or the
// smallest by the sign of gamma): ext = 2 planes (value, first index) of (b, c, groups)
__global__ void __launch_bounds__(256)
pool_from_extrema_kernel(int c, int groups, long long total, const float *__restrict__ ext,
                         const float *__restrict__ scale, const float *__restrict__ shift,
                         float *__restrict__ pooled, int *__restrict__ argmax,
                         float *__restrict__ ymax) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ch = (int)((i / groups) % c);
  const float yw = ext[i];
  pooled[i] = fmaxf(yw * scale[ch] + shift[ch], 0.f);
  argmax[i] = reinterpret_cast<const int *>(ext)[total + i];
  ymax[i] = yw;
}

// dgamma = s2, dbeta = s1, and the per-channel coefficients of dy (training mode); one wave
// per channel
struct BwdFinalize {
  int *tickets;  // one counter per channel, zero between launches (the caller's array)
  double count;
  int training;
  const float *gamma, *invstd;
  float *dgamma, *dbeta, *coef;
};

__device__ __forceinline__ void bwd_finalize_channel(int ch, int parts, const float *__restrict__ partial,
                                                     const BwdFinalize &f) {
  const int lane = lane_id();
  double s1 = 0.0, s2 = 0.0;
  const float *p = partial + (size_t)ch * parts * 2;
  for (int q = lane; q < parts; q += kWave) { s1 += coherent_load(p + q * 2); s2 += coherent_load(p + q * 2 + 1); }
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    s1 += shfl_xor_f64(s1, off);
    s2 += shfl_xor_f64(s2, off);
  }
  if (lane != 0) return;
  f.dbeta[ch] = (float)s1;
  f.dgamma[ch] = (float)s2;
  f.coef[ch * 3 + 0] = f.gamma[ch] * f.invstd[ch];
  // eval mode: statistics are constants, dy = gamma*invstd*dzh
  f.coef[ch * 3 + 1] = f.training ? (float)(s1 / f.count) : 0.f;
  f.coef[ch * 3 + 2] = f.training ? (float)(s2 / f.count) : 0.f;
}

__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(int c, int parts, const float *__restrict__ partial, BwdFinalize f) {
  const int ch = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
  if (ch >= c) return;
  bwd_finalize_channel(ch, parts, partial, f);
}

// sums (+ coefficients) of the BatchNorm + ReLU backward, and with APPLY also dy, per channel
template <bool APPLY>
__global__ void __launch_bounds__(kChThreads)
bn_channel_bwd_kernel(int bn, int c, int r, const float *__restrict__ y, const float *__restrict__ dz,
                      const float *__restrict__ scale, const float *__restrict__ shift,
                      const float *__restrict__ mean, const float *__restrict__ invstd,
                      float *__restrict__ dy, BwdFinalize f) {
  __shared__ double red[kChThreads / kWave][2];
  const int ch = blockIdx.x, nv = r >> 2, total = bn * nv;
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  float4 yy[4], gg[4];
  size_t at[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = threadIdx.x + u * kChThreads;
    yy[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    gg[u] = yy[u];
    at[u] = 0;
    if (i < total) {
      const int b = i / nv, j = i - b * nv;
      at[u] = ((size_t)b * c + ch) * r + (size_t)j * 4;
      yy[u] = *reinterpret_cast<const float4 *>(y + at[u]);
      gg[u] = *reinterpret_cast<const float4 *>(dz + at[u]);
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {  // (lanes beyond the channel hold zeros: dz = 0 adds nothing)
    gg[u].x = (yy[u].x * sc + sh > 0.f) ? gg[u].x : 0.f; gg[u].y = (yy[u].y * sc + sh > 0.f) ? gg[u].y : 0.f;
    gg[u].z = (yy[u].z * sc + sh > 0.f) ? gg[u].z : 0.f; gg[u].w = (yy[u].w * sc + sh > 0.f) ? gg[u].w : 0.f;
    s1 += (gg[u].x + gg[u].y) + (gg[u].z + gg[u].w);
    s2 += (gg[u].x * ((yy[u].x - mu) * is) + gg[u].y * ((yy[u].y - mu) * is)) +
          (gg[u].z * ((yy[u].z - mu) * is) + gg[u].w * ((yy[u].w - mu) * is));
  }
  double d1 = (double)s1, d2 = (double)s2;
  channel_sum2(d1, d2, red);
  const float a = f.gamma[ch] * f.invstd[ch];
  // eval mode: statistics are constants, dy = gamma*invstd*dzh
  const float c1 = f.training ? (float)(d1 / f.count) : 0.f, c2 = f.training ? (float)(d2 / f.count) : 0.f;
  if (threadIdx.x == 0) {
    f.dbeta[ch] = (float)d1;
    f.dgamma[ch] = (float)d2;
    f.coef[ch * 3 + 0] = a;
    f.coef[ch * 3 + 1] = c1;
    f.coef[ch * 3 + 2] = c2;
  }
  if constexpr (APPLY) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = threadIdx.x + u * kChThreads;
      if (i < total) {
        float4 o;
        o.x = a * (gg[u].x - c1 - ((yy[u].x - mu) * is) * c2);
        o.y = a * (gg[u].y - c1 - ((yy[u].y - mu) * is) * c2);
        o.z = a * (gg[u].z - c1 - ((yy[u].z - mu) * is) * c2);
        o.w = a * (gg[u].w - c1 - ((yy[u].w - mu) * is) * c2);
        *reinterpret_cast<float4 *>(dy + at[u]) = o;
      }
    }
  }
}

// ---- backward sums: s1 = sum dzh, s2 = sum dzh * xhat, dzh = dz * [y*scale+shift > 0] --------
__global__ void __launch_bounds__(kBnThreads)
bn_relu_bwd_partial_kernel(int c, int r, int slices, const float *__restrict__ y,
                           const float *__restrict__ dz, const float *__restrict__ scale,
                           const float *__restrict__ shift, const float *__restrict__ mean,
                           const float *__restrict__ invstd, float *__restrict__ partial,
                           BwdFinalize fin) {
  __shared__ float scratch[2 * kBnThreads / kWave];
  const int s = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int per = (r + slices - 1) / slices;
  const int lo = s * per, hi = lo + per < r ? lo + per : r;
  const size_t base = ((size_t)b * c + ch) * r;
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  float s1 = 0.f, s2 = 0.f;
  if (((r | lo | hi) & 3) == 0) {  // 16-byte loads of both tensors
    const float4 *vy = reinterpret_cast<const float4 *>(y + base + lo);
    const float4 *vd = reinterpret_cast<const float4 *>(dz + base + lo);
    const int nv = (hi - lo) >> 2;
    auto add = [&](const float4 &yy, const float4 &dd) {
      const float g0 = (yy.x * sc + sh > 0.f) ? dd.x : 0.f, g1 = (yy.y * sc + sh > 0.f) ? dd.y : 0.f;
      const float g2 = (yy.z * sc + sh > 0.f) ? dd.z : 0.f, g3 = (yy.w * sc + sh > 0.f) ? dd.w : 0.f;
      s1 += (g0 + g1) + (g2 + g3);
      s2 += (g0 * ((yy.x - mu) * is) + g1 * ((yy.y - mu) * is)) +
            (g2 * ((yy.z - mu) * is) + g3 * ((yy.w - mu) * is));
    };
    // four steps' loads in flight (one pair per step left a lane waiting out a memory round trip
    // per 32 bytes: 14 us per launch for 20 launches a step); the sums keep their order
    int i = threadIdx.x;
    for (; i + 3 * kBnThreads < nv; i += 4 * kBnThreads) {
      const float4 y0 = vy[i], d0 = vd[i];
      const float4 y1 = vy[i + kBnThreads], d1 = vd[i + kBnThreads];
      const float4 y2 = vy[i + 2 * kBnThreads], d2 = vd[i + 2 * kBnThreads];
      const float4 y3 = vy[i + 3 * kBnThreads], d3 = vd[i + 3 * kBnThreads];
      add(y0, d0); add(y1, d1); add(y2, d2); add(y3, d3);
    }
    for (; i < nv; i += kBnThreads) add(vy[i], vd[i]);
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += kBnThreads) {
      const float yy = y[base + i];
      const float g = (yy * sc + sh > 0.f) ? dz[base + i] : 0.f;
      s1 += g;
      s2 += g * ((yy - mu) * is);
    }
  }
  block_sum2(s1, s2, scratch);
  if (threadIdx.x == 0) {
    float *out = partial + ((size_t)ch * gridDim.z * slices + (size_t)b * slices + s) * 2;
    coherent_store(out, s1);
    coherent_store(out + 1, s2);
  }
  if (last_arrival(&fin.tickets[ch], (int)gridDim.z * slices) && threadIdx.x < kWave)
    bwd_finalize_channel(ch, (int)gridDim.z * slices, partial, fin);
}

// pooled layer: the same sums from the (B,C,m) tensors (dz is non-zero only at the arg-max)
__global__ void __launch_bounds__(kBnThreads)
pool_bwd_partial_kernel(int c, int m, const float *__restrict__ dpooled,
                        const float *__restrict__ ymax, const float *__restrict__ scale,
                        const float *__restrict__ shift, const float *__restrict__ mean,
                        const float *__restrict__ invstd, float *__restrict__ partial,
                        BwdFinalize fin) {
  __shared__ float scratch[2 * kBnThreads / kWave];
  const int ch = blockIdx.y, b = blockIdx.z;
  const size_t base = ((size_t)b * c + ch) * m;
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < m; i += kBnThreads) {
    const float yy = ymax[base + i];
    const float g = (yy * sc + sh > 0.f) ? dpooled[base + i] : 0.f;
    s1 += g;
    s2 += g * ((yy - mu) * is);
  }
  block_sum2(s1, s2, scratch);
  if (threadIdx.x == 0) {
    float *out = partial + ((size_t)ch * gridDim.z + b) * 2;
    coherent_store(out, s1);
    coherent_store(out + 1, s2);
  }
  if (last_arrival(&fin.tickets[ch], (int)gridDim.z) && threadIdx.x < kWave)
    bwd_finalize_channel(ch, (int)gridDim.z, partial, fin);
}

// dy = a * (dzh - c1 - xhat*c2)
template <bool VEC>
__global__ void __launch_bounds__(256)
bn_relu_bwd_apply_kernel(int c, int r, const float *__restrict__ y, const float *__restrict__ dz,
                         const float *__restrict__ scale, const float *__restrict__ shift,
                         const float *__restrict__ mean, const float *__restrict__ invstd,
                         const float *__restrict__ coef, float *__restrict__ dy) {
  const int ch = blockIdx.y, b = blockIdx.z;
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  const float a = coef[ch * 3], c1 = coef[ch * 3 + 1], c2 = coef[ch * 3 + 2];
  const size_t base = ((size_t)b * c + ch) * r;
  if (VEC) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= r) return;
    const float4 yy = *reinterpret_cast<const float4 *>(y + base + i);
    const float4 g = *reinterpret_cast<const float4 *>(dz + base + i);
    float4 o;
    o.x = a * (((yy.x * sc + sh > 0.f) ? g.x : 0.f) - c1 - ((yy.x - mu) * is) * c2);
    o.y = a * (((yy.y * sc + sh > 0.f) ? g.y : 0.f) - c1 - ((yy.y - mu) * is) * c2);
    o.z = a * (((yy.z * sc + sh > 0.f) ? g.z : 0.f) - c1 - ((yy.z - mu) * is) * c2);
    o.w = a * (((yy.w * sc + sh > 0.f) ? g.w : 0.f) - c1 - ((yy.w - mu) * is) * c2);
    *reinterpret_cast<float4 *>(dy + base + i) = o;
  } else {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < r) {
      const float yy = y[base + i];
      dy[base + i] = a * (((yy * sc + sh > 0.f) ? dz[base + i] : 0.f) - c1 - ((yy - mu) * is) * c2);
    }
  }
}

// pooled layer: dy[b,c,j,k] = a * ([k == argmax && relu'] * dpooled - c1 - xhat*c2)
template <bool VEC>
__global__ void __launch_bounds__(256)
pool_bwd_apply_kernel(int c, int m, int ns, const float *__restrict__ y,
                      const float *__restrict__ dpooled, const int *__restrict__ argmax,
                      const float *__restrict__ scale, const float *__restrict__ shift,
                      const float *__restrict__ mean, const float *__restrict__ invstd,
                      const float *__restrict__ coef, float *__restrict__ dy) {
  const int ch = blockIdx.y, b = blockIdx.z;
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  const float a = coef[ch * 3], c1 = coef[ch * 3 + 1], c2 = coef[ch * 3 + 2];
  const int r = m * ns;
  const size_t base = ((size_t)b * c + ch) * r;
  const size_t pbase = ((size_t)b * c + ch) * m;
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1);
  if (i0 >= r) return;
  float out[4];
#pragma unroll
  for (int t = 0; t < (VEC ? 4 : 1); ++t) {
    const int i = i0 + t;
    const int row = i / ns, k = i - row * ns;
    const float yy = y[base + i];
    float g = 0.f;
    if (argmax[pbase + row] == k && yy * sc + sh > 0.f) g = dpooled[pbase + row];
    out[t] = a * (g - c1 - ((yy - mu) * is) * c2);
  }
  if (VEC) *reinterpret_cast<float4 *>(dy + base + i0) = make_float4(out[0], out[1], out[2], out[3]);
  else dy[base + i0] = out[0];
}

int slices_for(int r) {
  int s = r / 8192;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return s;
}

}  // namespace


#define MLP_API extern "C" __attribute__((visibility("default")))

// number of floats of scratch the statistics kernels need
MLP_API size_t mlp_bn_workspace_floats(int b, int c, int r) {
  return (size_t)c * b * slices_for(r) * 3;
}

// Training-mode forward statistics + coefficients.  running_* may be NULL.
MLP_API int mlp_bn_train_stats(int b, int c, int r, const float *y, const float *gamma,
                               const float *beta, float eps, float momentum, float *running_mean,
                               float *running_var, float *mean, float *invstd, float *scale,
                               float *shift, float *workspace, int *tickets, void *stream_) {
  if (b <= 0 || c <= 0 || r <= 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  const int slices = slices_for(r);
  const FwdFinalize fin = {tickets, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift};
  if (fin.tickets == nullptr) return (int)hipErrorInvalidValue;
  if (channel_form(b, c, r, y, nullptr, nullptr)) {
    hipLaunchKernelGGL(bn_channel_stats_kernel, dim3(c), dim3(kChThreads), 0, stream, b, c, r, y, fin);
    return pn2_launch_status();
  }
  hipLaunchKernelGGL(bn_partial_stats_kernel, dim3(slices, c, b), dim3(kBnThreads), 0, stream, c,
                     r, slices, y, workspace, fin);
  return pn2_launch_status();
}

// Training-mode coefficients from the (mean, M2) pairs the forward GEMM left behind
// (mlp_gemm_forward_stats): the statistics pass over y is gone.  scratch: device memory of
// mlp_bn_finalize_pairs_scratch_bytes(c) bytes.
MLP_API size_t mlp_bn_finalize_pairs_scratch_bytes(int c) {
  return sizeof(double) * 3 * (size_t)kPairSlices * (size_t)(c > 0 ? c : 0);
}

MLP_API int mlp_bn_finalize_pairs(int c, int parts, int n_part, const float *pairs,
                                  const float *gamma, const float *beta, float eps, float momentum,
                                  float *running_mean, float *running_var, float *mean,
                                  float *invstd, float *scale, float *shift, void *scratch,
                                  void *stream_) {
  if (c <= 0 || parts <= 0 || n_part <= 0) return 0;
  if (!scratch) return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  double *sums = reinterpret_cast<double *>(scratch);
  const FwdFinalize fin = {nullptr, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift};
  // (MLP_BN_PAIRS_ONE_LAUNCH = the largest number of parts finished in one launch; 0: always two)
  static const int one_limit = getenv("MLP_BN_PAIRS_ONE_LAUNCH") ? atoi(getenv("MLP_BN_PAIRS_ONE_LAUNCH")) : kPairsOneLaunch;
  if (parts <= one_limit) {
    hipLaunchKernelGGL(bn_pairs_one_kernel, dim3(pn2_ceil_div(c, kWave)), dim3(1024), 0, stream, c, parts, n_part,
                       pairs, fin);
    return pn2_launch_status();
  }
  hipLaunchKernelGGL(bn_pairs_stage1_kernel, dim3(pn2_ceil_div(c, kWave), kPairSlices), dim3(1024),
                     0, stream, c, parts, pairs, sums);
  hipLaunchKernelGGL(bn_pairs_stage2_kernel, dim3(pn2_ceil_div(c, 256)), dim3(256), 0, stream, c,
                     parts, n_part, pairs, sums, fin);
  return pn2_launch_status();
}

MLP_API int mlp_bn_eval_coeff(int c, const float *gamma, const float *beta, float eps,
                              const float *running_mean, const float *running_var, float *mean,
                              float *invstd, float *scale, float *shift, void *stream_) {
  if (c <= 0) return 0;
  hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(pn2_ceil_div(c, 256)), dim3(256), 0,
                     (hipStream_t)stream_, c, gamma, beta, eps, running_mean, running_var, mean,
                     invstd, scale, shift);
  return pn2_launch_status();
}

MLP_API int mlp_bn_relu_apply(int b, int c, int r, const float *y, const float *scale,
                              const float *shift, float *z, void *stream_) {
  if (b <= 0 || c <= 0 || r <= 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  if (r % 4 == 0)
    hipLaunchKernelGGL(bn_relu_apply_kernel<true>, dim3(pn2_ceil_div(r, 1024), c, b), dim3(256), 0,
                       stream, c, r, y, scale, shift, z);
  else
    hipLaunchKernelGGL(bn_relu_apply_kernel<false>, dim3(pn2_ceil_div(r, 256), c, b), dim3(256), 0,
                       stream, c, r, y, scale, shift, z);
  return pn2_launch_status();
}

MLP_API int mlp_bn_relu_pool(int b, int c, int m, int ns, const float *y, const float *scale,
                             const float *shift, float *pooled, int *argmax, float *ymax,
                             void *stream_) {
  if (b <= 0 || c <= 0 || m <= 0 || ns <= 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
#define POOL(LPR)                                                                              \
  hipLaunchKernelGGL(bn_relu_pool_kernel<LPR>, dim3(pn2_ceil_div((long long)m * LPR, 256), c, b), \
                     dim3(256), 0, stream, c, m, y, scale, shift, pooled, argmax, ymax)
  switch (ns) {
    case 4: POOL(1); break;
    case 8: POOL(2); break;
    case 16: POOL(4); break;
    case 32: POOL(8); break;
    case 64: POOL(16); break;
    default:
      hipLaunchKernelGGL(bn_relu_pool_generic_kernel, dim3(pn2_ceil_div(m, 256), c, b), dim3(256),
                         0, stream, c, m, ns, y, scale, shift, pooled, argmax, ymax);
  }
#undef POOL
  return pn2_launch_status();
}

// dz (b,c,r) -> dy (b,c,r), dgamma, dbeta.  coef: 3*c floats of scratch; workspace as above.
MLP_API int mlp_bn_relu_backward(int b, int c, int r, int training, const float *y,
                                 const float *dz, const float *gamma, const float *scale,
                                 const float *shift, const float *mean, const float *invstd,
                                 float *dy, float *dgamma, float *dbeta, float *coef,
                                 float *workspace, int *tickets, void *stream_) {
  if (b <= 0 || c <= 0 || r <= 0) return 0;
  if (tickets == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  if (channel_form(b, c, r, y, dz, dy)) {  // sums and dy in one launch, y and dz read once
    hipLaunchKernelGGL(bn_channel_bwd_kernel<true>, dim3(c), dim3(kChThreads), 0, stream, b, c, r, y, dz, scale,
                       shift, mean, invstd, dy,
                       BwdFinalize{tickets, (double)b * (double)r, training, gamma, invstd, dgamma, dbeta, coef});
    return pn2_launch_status();
  }
  const int slices = slices_for(r);
  hipLaunchKernelGGL(bn_relu_bwd_partial_kernel, dim3(slices, c, b), dim3(kBnThreads), 0, stream,
                     c, r, slices, y, dz, scale, shift, mean, invstd, workspace,
                     BwdFinalize{tickets, (double)b * (double)r, training, gamma, invstd, dgamma, dbeta, coef});
  if (r % 4 == 0)
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<true>, dim3(pn2_ceil_div(r, 1024), c, b), dim3(256),
                       0, stream, c, r, y, dz, scale, shift, mean, invstd, coef, dy);
  else
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<false>, dim3(pn2_ceil_div(r, 256), c, b), dim3(256),
                       0, stream, c, r, y, dz, scale, shift, mean, invstd, coef, dy);
  return pn2_launch_status();
}

// sums + coefficients only (the dy tensor is formed inside the GEMM operand loads instead)
MLP_API int mlp_bn_relu_backward_stats(int b, int c, int r, int training, const float *y,
                                       const float *dz, const float *gamma, const float *scale,
                                       const float *shift, const float *mean, const float *invstd,
                                       float *dgamma, float *dbeta, float *coef, float *workspace,
                                       int *tickets, void *stream_) {
  if (b <= 0 || c <= 0 || r <= 0) return 0;
  if (tickets == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  if (channel_form(b, c, r, y, dz, nullptr)) {
    hipLaunchKernelGGL(bn_channel_bwd_kernel<false>, dim3(c), dim3(kChThreads), 0, stream, b, c, r, y, dz, scale,
                       shift, mean, invstd, nullptr,
                       BwdFinalize{tickets, (double)b * (double)r, training, gamma, invstd, dgamma, dbeta, coef});
    return pn2_launch_status();
  }
  const int slices = slices_for(r);
  hipLaunchKernelGGL(bn_relu_bwd_partial_kernel, dim3(slices, c, b), dim3(kBnThreads), 0, stream,
                     c, r, slices, y, dz, scale, shift, mean, invstd, workspace,
                     BwdFinalize{tickets, (double)b * (double)r, training, gamma, invstd, dgamma, dbeta, coef});
  return pn2_launch_status();
}

// the same coefficients from (s1, s2) partials that another kernel left behind (the fused
// backward GEMM of the layer above: mlp_gemm_backward_fused): partial (c, parts, 2)
MLP_API int mlp_bn_backward_finalize(int c, int parts, double count, int training,
                                     const float *partial, const float *gamma,
                                     const float *invstd, float *dgamma, float *dbeta, float *coef,
                                     void *stream_) {
  if (c <= 0 || parts <= 0) return 0;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(pn2_ceil_div(c, 256 / kWave)), dim3(256), 0,
                     (hipStream_t)stream_, c, parts, partial,
                     BwdFinalize{nullptr, count, training, gamma, invstd, dgamma, dbeta, coef});
  return pn2_launch_status();
}

// (pooled, argmax, ymax) of mlp_bn_relu_pool from the extrema planes of mlp_gemm_forward_stats_pool
MLP_API int mlp_bn_pool_from_extrema(int b, int c, int groups, const float *ext, const float *scale,
                                     const float *shift, float *pooled, int *argmax, float *ymax,
                                     void *stream_) {
  if (b <= 0 || c <= 0 || groups <= 0) return 0;
  const long long total = (long long)b * c * groups;
  hipLaunchKernelGGL(pool_from_extrema_kernel, dim3(pn2_ceil_div(total, 256)), dim3(256), 0,
                     (hipStream_t)stream_, c, groups, total, ext, scale, shift, pooled, argmax, ymax);
  return pn2_launch_status();
}

// pooled layer backward: dpooled (b,c,m) -> dy (b,c,m,ns)
MLP_API int mlp_bn_relu_pool_backward(int b, int c, int m, int ns, int training, const float *y,
                                      const float *dpooled, const int *argmax, const float *ymax,
                                      const float *gamma, const float *scale, const float *shift,
                                      const float *mean, const float *invstd, float *dy,
                                      float *dgamma, float *dbeta, float *coef, float *workspace,
                                      int *tickets, void *stream_) {
  if (b <= 0 || c <= 0 || m <= 0 || ns <= 0) return 0;
  if (tickets == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  // the same sums over the pooled tensors: (ymax, dpooled) in the roles of (y, dz), m values per cloud
  // and channel -- one workgroup per channel where they fit (bn_channel_bwd_kernel)
  if (channel_form(b, c, m, ymax, dpooled, nullptr))
    hipLaunchKernelGGL(bn_channel_bwd_kernel<false>, dim3(c), dim3(kChThreads), 0, stream, b, c, m, ymax, dpooled,
                       scale, shift, mean, invstd, nullptr,
                       BwdFinalize{tickets, (double)b * (double)m * (double)ns, training, gamma, invstd,
                                   dgamma, dbeta, coef});
  else
  hipLaunchKernelGGL(pool_bwd_partial_kernel, dim3(1, c, b), dim3(kBnThreads), 0, stream, c, m,
                     dpooled, ymax, scale, shift, mean, invstd, workspace,
                     BwdFinalize{tickets, (double)b * (double)m * (double)ns, training, gamma, invstd,
                                 dgamma, dbeta, coef});
  if (dy == nullptr) return pn2_launch_status();  // statistics only (mlp_gemm_*_pooled form dy)
  const long long r = (long long)m * ns;
  if (r % 4 == 0)
    hipLaunchKernelGGL(pool_bwd_apply_kernel<true>, dim3(pn2_ceil_div(r, 1024), c, b), dim3(256), 0,
                       stream, c, m, ns, y, dpooled, argmax, scale, shift, mean, invstd, coef, dy);
  else
    hipLaunchKernelGGL(pool_bwd_apply_kernel<false>, dim3(pn2_ceil_div(r, 256), c, b), dim3(256), 0,
                       stream, c, m, ns, y, dpooled, argmax, scale, shift, mean, invstd, coef, dy);
  return pn2_launch_status();
}
