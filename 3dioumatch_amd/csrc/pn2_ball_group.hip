// 3dioumatch_amd/csrc/pn2_ball_group.hip -- ball query + neighbourhood gather/scatter, gfx950.
//
// Semantics: reference ball_query_gpu.cu:14-49 (K4), group_points_gpu.cu:13-33 (K5),
// :48-69 (K6); restated in SURVEY App. A.3/A.4.  The reference launches ONE block per
// cloud and lets each thread scan all N points serially with stride-3 loads.
//
// Design here (brute-force tier; the cell-list tier for large clouds lives in
// pn2_ball_grid.hip):
//  * one 64-lane wavefront owns QW centroids (coordinates wave-uniform), lanes own 64
//    CONSECUTIVE points per step (coalesced loads, points reused for all QW centroids);
//  * hits are compacted IN INDEX ORDER with a wave ballot + prefix popcount, so the
//    "first nsample indices, ascending" contract holds by construction and the early exit
//    (cnt == nsample) is wave-uniform -- no lane ever idles waiting for a slower centroid;
//  * the reference's "pre-fill the row with the first hit" (ball_query_gpu.cu:39-43) becomes
//    a tail fill after the scan; rows without a hit are written as zeros, so the output
//    needs no pre-zeroing.
#include "common.h"
#include <stdlib.h>
#include "ball_common.h"

namespace {

template <int QW>
__global__ void __launch_bounds__(256)
ball_query_bf_kernel(int n, int m, float radius2, int nsample,
                     const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                     int *__restrict__ idx) {
  const BlockId blk = xcd_block_id();
  const int b = blk.y;
  const int wave = threadIdx.x / kWave;
  const int j0 = (blk.x * (256 / kWave) + wave) * QW;
  if (j0 >= m) return;
  ball_query_wave_scan<QW>(xyz + (size_t)b * n * 3, n, new_xyz + ((size_t)b * m + j0) * 3,
                           m - j0 < QW ? m - j0 : QW, radius2, nsample,
                           idx + ((size_t)b * m + j0) * nsample);
}

// out[b,l,e] = points[b,l,idx[b,e]],  e over npoints*nsample  (group_points_gpu.cu:13-33).
// Each lane owns 4 consecutive e (16-byte idx load / 16-byte stores when VEC) and walks
// CPT... channels, so an index is fetched once per channel group.
template <bool VEC>
__global__ void __launch_bounds__(256)
group_points_kernel(int c, int n, int mns, const float *__restrict__ points,
                    const int *__restrict__ idx, float *__restrict__ out) {
  const BlockId blk = xcd_block_id();
  const int b = blk.z;
  const int e = (blk.x * 256 + threadIdx.x) * 4;
  if (e >= mns) return;
  const int *ib = idx + (size_t)b * mns + e;
  int i0, i1 = 0, i2 = 0, i3 = 0;
  const int live = mns - e < 4 ? mns - e : 4;
  if (VEC) {
    const int4 v = *reinterpret_cast<const int4 *>(ib);
    i0 = v.x; i1 = v.y; i2 = v.z; i3 = v.w;
  } else {
    i0 = ib[0];
    if (live > 1) i1 = ib[1];
    if (live > 2) i2 = ib[2];
    if (live > 3) i3 = ib[3];
  }
  for (int l = blk.y; l < c; l += gridDim.y) {
    const float *src = points + ((size_t)b * c + l) * n;
    float *dst = out + ((size_t)b * c + l) * mns + e;
    if (VEC) {
      float4 v;
      v.x = src[i0]; v.y = src[i1]; v.z = src[i2]; v.w = src[i3];
      *reinterpret_cast<float4 *>(dst) = v;
    } else {
      dst[0] = src[i0];
      if (live > 1) dst[1] = src[i1];
      if (live > 2) dst[2] = src[i2];
      if (live > 3) dst[3] = src[i3];
    }
  }
}

// LDS-staged gather.  A 4-byte gather whose 64 lanes fall in 64 different cache lines costs the
// CU's address unit ~64 cycles, and a ball's indices are scattered over the whole cloud: the
// kernel above is bound by that (5 us per channel at N = 40000, 2048 x 64 indices, for 8.4 MB of
// traffic).  Here workgroup (slice, channel, cloud) first copies its channel row -- at most LDSF
// floats; 40960 = the CU's whole 160 KB LDS -- with coalesced 16-byte loads (an L2 hit on the
// cloud's XCD), then serves its slice of the index array from LDS, where a scattered 64-lane
// read is a handful of bank-conflict cycles.
template <int LDSF, bool VEC>
__global__ void __launch_bounds__(1024)
group_points_lds_kernel(int c, int n, int mns, const float *__restrict__ points,
                        const int *__restrict__ idx, float *__restrict__ out) {
  __shared__ float row[LDSF];
  const BlockId blk = xcd_block_id();
  const int b = blk.z, l = blk.y;
  const float *src = points + ((size_t)b * c + l) * n;
  if ((n & 3) == 0 && n > 0) {
    // all of a lane's 16-byte loads are issued before the first LDS write (one L2 round trip
    // for the whole row instead of one per 16 KB)
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *r4 = reinterpret_cast<float4 *>(row);
    constexpr int kIter = LDSF / 4096;
    float4 v[kIter];
    const int n4 = n / 4;  // >= 1 here
#pragma unroll
    for (int u = 0; u < kIter; ++u) {  // unconditional (clamped) so that the loads stay batched
      const int t = threadIdx.x + u * 1024;
      v[u] = s4[t < n4 ? t : n4 - 1];
    }
#pragma unroll
    for (int u = 0; u < kIter; ++u)  // keeps the compiler from sinking each load to its write
      asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
    for (int u = 0; u < kIter; ++u)
      if (threadIdx.x + u * 1024 < n4) r4[threadIdx.x + u * 1024] = v[u];
  } else {
    for (int t = threadIdx.x; t < n; t += 1024) row[t] = src[t];
  }
  __syncthreads();
  const int chunks = (mns + 3) / 4;
  const int per = (chunks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int q1 = (blk.x + 1) * per < chunks ? (blk.x + 1) * per : chunks;
  const int *ib = idx + (size_t)b * mns;
  float *dst = out + ((size_t)b * c + l) * mns;
  for (int q = blk.x * per + threadIdx.x; q < q1; q += 1024) {
    if (VEC) {
      const int4 v = reinterpret_cast<const int4 *>(ib)[q];
      float4 o;
      o.x = row[v.x]; o.y = row[v.y]; o.z = row[v.z]; o.w = row[v.w];
      reinterpret_cast<float4 *>(dst)[q] = o;
    } else {
      for (int e = q * 4; e < q * 4 + 4 && e < mns; ++e) dst[e] = row[ib[e]];
    }
  }
}

// grad_points[b,l,idx[b,e]] += grad_out[b,l,e]   (group_points_gpu.cu:48-69)
template <bool VEC>
__global__ void __launch_bounds__(256)
group_points_grad_kernel(int c, int n, int mns, const float *__restrict__ grad_out,
                         const int *__restrict__ idx, float *__restrict__ grad_points) {
  const BlockId blk = xcd_block_id();
  const int b = blk.z;
  const int e = (blk.x * 256 + threadIdx.x) * 4;
  if (e >= mns) return;
  const int *ib = idx + (size_t)b * mns + e;
  int i0, i1 = 0, i2 = 0, i3 = 0;
  const int live = mns - e < 4 ? mns - e : 4;
  if (VEC) {
    const int4 v = *reinterpret_cast<const int4 *>(ib);
    i0 = v.x; i1 = v.y; i2 = v.z; i3 = v.w;
  } else {
    i0 = ib[0];
    if (live > 1) i1 = ib[1];
    if (live > 2) i2 = ib[2];
    if (live > 3) i3 = ib[3];
  }
  for (int l = blk.y; l < c; l += gridDim.y) {
    float *dst = grad_points + ((size_t)b * c + l) * n;
    const float *src = grad_out + ((size_t)b * c + l) * mns + e;
    if (VEC) {
      const float4 g = *reinterpret_cast<const float4 *>(src);
      // first-hit padding repeats one index many times in a row: merge equal neighbours
      // before touching memory (fewer same-address atomics)
      float a0 = g.x, a1 = g.y, a2 = g.z, a3 = g.w;
      if (i1 == i0) { a1 = __fadd_rn(a0, a1); a0 = 0.f; }
      if (i2 == i1) { a2 = __fadd_rn(a1, a2); a1 = 0.f; }
      if (i3 == i2) { a3 = __fadd_rn(a2, a3); a2 = 0.f; }
      if (i1 != i0) atomicAdd(dst + i0, a0);
      if (i2 != i1) atomicAdd(dst + i1, a1);
      if (i3 != i2) atomicAdd(dst + i2, a2);
      atomicAdd(dst + i3, a3);
    } else {
      atomicAdd(dst + i0, src[0]);
      if (live > 1) atomicAdd(dst + i1, src[1]);
      if (live > 2) atomicAdd(dst + i2, src[2]);
      if (live > 3) atomicAdd(dst + i3, src[3]);
    }
  }
}

// Scatter-add with the destination rows privatised in LDS (n*4 bytes per channel): one
// workgroup owns CPW channels of one cloud, streams their grad_out rows once (coalesced
// 16-byte loads), accumulates with LDS atomics (ds_add_f32) and writes each (b,c,:) row back
// exactly once -- no global atomics, no pre-zeroing.  Used whenever a row fits (n <= 16384);
// the global-atomic kernel above remains for huge n (SA1, where C <= 4 anyway).
template <int CPW, bool VEC>
__global__ void __launch_bounds__(1024)
group_points_grad_lds_kernel(int c, int n, int mns, const float *__restrict__ grad_out,
                             const int *__restrict__ idx, float *__restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float acc[];
  const BlockId blk = xcd_block_id();
  const int b = blk.y;
  const int l0 = blk.x * CPW;
  const int nc = c - l0 < CPW ? c - l0 : CPW;
  for (int t = threadIdx.x; t < nc * n; t += 1024) acc[t] = 0.f;
  __syncthreads();
  const int *ib = idx + (size_t)b * mns;
  const float *src = grad_out + ((size_t)b * c + l0) * mns;
  for (int e = threadIdx.x * 4; e < mns; e += 1024 * 4) {
    int i0, i1 = 0, i2 = 0, i3 = 0;
    const int live = mns - e < 4 ? mns - e : 4;
    if (VEC) {
      const int4 v = *reinterpret_cast<const int4 *>(ib + e);
      i0 = v.x; i1 = v.y; i2 = v.z; i3 = v.w;
    } else {
      i0 = ib[e];
      if (live > 1) i1 = ib[e + 1];
      if (live > 2) i2 = ib[e + 2];
      if (live > 3) i3 = ib[e + 3];
    }
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
      if (cc < nc) {
        float *row = acc + cc * n;
        const float *g = src + (size_t)cc * mns + e;
        if (VEC) {
          const float4 v = *reinterpret_cast<const float4 *>(g);
          float a0 = v.x, a1 = v.y, a2 = v.z, a3 = v.w;
          if (i1 == i0) { a1 = __fadd_rn(a0, a1); } else { atomicAdd(row + i0, a0); }
          if (i2 == i1) { a2 = __fadd_rn(a1, a2); } else { atomicAdd(row + i1, a1); }
          if (i3 == i2) { a3 = __fadd_rn(a2, a3); } else { atomicAdd(row + i2, a2); }
          atomicAdd(row + i3, a3);
        } else {
          atomicAdd(row + i0, g[0]);
          if (live > 1) atomicAdd(row + i1, g[1]);
          if (live > 2) atomicAdd(row + i2, g[2]);
          if (live > 3) atomicAdd(row + i3, g[3]);
        }
      }
    }
  }
  __syncthreads();
  float *dst = grad_points + ((size_t)b * c + l0) * n;
  for (int t = threadIdx.x; t < nc * n; t += 1024) dst[t] = acc[t];
}

// The same with ONE destination row in the CU's whole LDS (n <= 40960 floats = 160 KB): SA1-sized
// clouds, where the global-atomic kernel ran at 135 GB/s (every add a memory-side atomic) and
// needed a zero-fill pass.  A row's indices are re-read by each of its channels' workgroups
// (an L2 hit on the cloud's XCD).
template <bool VEC>
__global__ void __launch_bounds__(1024)
group_points_grad_lds_big_kernel(int c, int n, int mns, const float *__restrict__ grad_out,
                                 const int *__restrict__ idx, float *__restrict__ grad_points) {
  __shared__ float acc[40960];
  const BlockId blk = xcd_block_id();
  const int b = blk.y, l = blk.x;
  for (int t = threadIdx.x; t < n; t += 1024) acc[t] = 0.f;
  __syncthreads();
  const int *ib = idx + (size_t)b * mns;
  const float *g = grad_out + ((size_t)b * c + l) * mns;
  if (VEC) {
    for (int e = threadIdx.x * 4; e < mns; e += 1024 * 4) {
      const int4 i = *reinterpret_cast<const int4 *>(ib + e);
      const float4 v = *reinterpret_cast<const float4 *>(g + e);
      // first-hit padding repeats one index many times in a row: merge equal neighbours
      float a1 = v.y, a2 = v.z, a3 = v.w;
      if (i.y == i.x) { a1 = __fadd_rn(v.x, a1); } else { atomicAdd(acc + i.x, v.x); }
      if (i.z == i.y) { a2 = __fadd_rn(a1, a2); } else { atomicAdd(acc + i.y, a1); }
      if (i.w == i.z) { a3 = __fadd_rn(a2, a3); } else { atomicAdd(acc + i.z, a2); }
      atomicAdd(acc + i.w, a3);
    }
  } else {
    for (int e = threadIdx.x; e < mns; e += 1024) atomicAdd(acc + ib[e], g[e]);
  }
  __syncthreads();
  float *dst = grad_points + ((size_t)b * c + l) * n;
  for (int t = threadIdx.x; t < n; t += 1024) dst[t] = acc[t];
}

// The same for FEW rows (b * c < 128 workgroups: SA1's xyz / height channels, C = 3 + 1): a row is
// cut into `slices` ranges of destination points, workgroup (slice, channel, cloud) keeps its range
// in LDS and scans the WHOLE index array for hits in it -- the index / gradient rows are read
// `slices` times from L2 (1 MB each at SA1) in exchange for `slices` times the workgroups and a
// fraction of the same-address pressure: 153 -> 34 us at B = 8, C = 4, n = 40 000, m*ns = 131 072.
// (Round 6, measured and not kept: all channels of a cloud in ONE workgroup per range, the index
// array scanned once in batches, the runs that hit the range compacted into an LDS list and added
// in a second phase with independent gradient loads -- 69 .. 100 us against this kernel's 50: the
// scan of the whole index array by 256-512 workgroups costs more vector instructions than the
// re-read gradient rows cost bandwidth, and a returning LDS atomic per lane on ONE address, the
// list's counter, alone took 67 us before it was aggregated per wave; profiles/r6_ops_time.json.)
template <bool VEC>
__global__ void __launch_bounds__(1024)
group_points_grad_lds_range_kernel(int c, int n, int mns, int slices, const float *__restrict__ grad_out,
                                   const int *__restrict__ idx, float *__restrict__ grad_points) {
  __shared__ float acc[10240];
  const BlockId blk = xcd_block_id();  // grid (slices * c, b): a cloud's workgroups share an XCD's L2
  const int b = blk.y, l = blk.x / slices, sl = blk.x % slices;
  const int per = (n + slices - 1) / slices;
  const int lo = sl * per;
  const unsigned len = (unsigned)((lo + per < n ? lo + per : n) - lo);
  for (int t = threadIdx.x; t < (int)len; t += 1024) acc[t] = 0.f;
  __syncthreads();
  const int *ib = idx + (size_t)b * mns;
  const float *g = grad_out + ((size_t)b * c + l) * mns;
  auto add = [&](int i, float v) {
    const unsigned o = (unsigned)(i - lo);
    if (o < len) atomicAdd(acc + o, v);
  };
  if (VEC) {
    for (int e = threadIdx.x * 4; e < mns; e += 1024 * 4) {
      const int4 i = *reinterpret_cast<const int4 *>(ib + e);
      const float4 v = *reinterpret_cast<const float4 *>(g + e);
      // first-hit padding repeats one index many times in a row: merge equal neighbours
      float a1 = v.y, a2 = v.z, a3 = v.w;
      if (i.y == i.x) { a1 = __fadd_rn(v.x, a1); } else { add(i.x, v.x); }
      if (i.z == i.y) { a2 = __fadd_rn(a1, a2); } else { add(i.y, a1); }
      if (i.w == i.z) { a3 = __fadd_rn(a2, a3); } else { add(i.z, a2); }
      add(i.w, a3);
    }
  } else {
    for (int e = threadIdx.x; e < mns; e += 1024) add(ib[e], g[e]);
  }
  __syncthreads();
  float *dst = grad_points + ((size_t)b * c + l) * n + lo;
  for (int t = threadIdx.x; t < (int)len; t += 1024) dst[t] = acc[t];
}

// ---- scatter-add through an inverse index (no float atomics per element) --------------------
// The LDS-privatised scatter-add above spends its time in ds_add_f32 (about 0.4 lane-adds per
// clock and CU: 128 us for SA2's 134 MB).  The index array is reused by every channel and every
// step's backward, so it is inverted ONCE (in the prefetched index chain): the m*ns positions of
// a cloud counting-sorted by the point they refer to, one packed entry (point << 16 | position)
// each.  The backward of a channel stages its gradient row in LDS; every lane owns CHUNK
// consecutive sorted entries, gathers their values from the row (plain LDS reads, 32 lanes per
// clock) and adds them up in a register; a run that starts and ends inside a lane is stored
// once, only the two runs a lane may share with its neighbours use an LDS atomic.
//
// Entry s of the sorted order lives at [(s % CHUNK) * 1024 + s / CHUNK] so that lane t's j-th
// entry is at [j * 1024 + t]: coalesced.  CHUNK = 4/8/16/32 (the smallest that covers m*ns),
// unused slots hold 0xFFFFFFFF.
constexpr int kInvMaxEntries = 32768;  // gradient row staged in 128 KB of LDS
constexpr int kInvMaxPoints = 4096;    // destination row / counters in 16 KB of LDS
constexpr unsigned kInvPad = 0xFFFFFFFFu;

int inverse_chunk(int mns) {
  int chunk = 4;
  while (chunk * 1024 < mns) chunk *= 2;
  return chunk;
}

// VEC: m*ns is a multiple of 4.  A lane then owns whole int4 quads of the index array -- ALL of
// its quads are loaded before the first LDS atomic (one trip to the L2 instead of one per pass
// and element) and equal neighbours inside a quad are counted and ranked as ONE run: first-hit
// padding repeats an index up to nsample times in a row, and every repeat was a same-address LDS
// atomic (32 us at SA2's 8 x 1024 x 32; the order of a point's entries is free -- the sorted
// backward adds them up whatever it is).
template <bool VEC>
__global__ void __launch_bounds__(1024)
group_inverse_kernel(int n, int mns, int chunk_log2, const int *__restrict__ idx,
                     unsigned *__restrict__ inv) {
  __shared__ int cnt[kInvMaxPoints];
  __shared__ int wave_tot[16];
  // the sorted entries are put together in LDS (scattered 4-byte writes) and leave as whole rows
  __shared__ unsigned stage[VEC ? kInvMaxEntries : 1];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int *ib = idx + (size_t)b * mns;
  unsigned *out = inv + ((size_t)b << (chunk_log2 + 10));
  const int chunk_mask = (1 << chunk_log2) - 1;
  constexpr int Q = kInvMaxEntries / 4096;  // quads per lane at most
  int4 v[Q];
  if (VEC) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int e = (q * 1024 + tid) * 4;
      v[q] = e < mns ? *reinterpret_cast<const int4 *>(ib + e) : make_int4(-1, -1, -1, -1);
    }
  }
  for (int t = tid; t < kInvMaxPoints; t += 1024) cnt[t] = 0;
  __syncthreads();
  // runs of a quad: (key, length) of the run that STARTS at element u, length 0 elsewhere
  auto runs = [](const int4 &q, int len[4]) {
    const int k[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int u = 3; u >= 0; --u)
      len[u] = 1 + ((u < 3 && k[u + 1] == k[u]) ? len[u + 1] : 0);
#pragma unroll
    for (int u = 3; u >= 1; --u)
      if (k[u] == k[u - 1]) len[u] = 0;
  };
  if (VEC) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (v[q].x < 0) continue;
      int len[4];
      runs(v[q], len);
      const int k[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (len[u] > 0) atomicAdd(&cnt[k[u]], len[u]);
    }
  } else {
    for (int e = tid; e < mns; e += 1024) atomicAdd(&cnt[ib[e]], 1);
  }
  __syncthreads();
  {  // exclusive scan of the counters, four per lane
    int c4[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { c4[q] = cnt[tid * 4 + q]; sum += c4[q]; }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int o = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) wave_tot[w] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int q = 0; q < w; ++q) run += wave_tot[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) { cnt[tid * 4 + q] = run; run += c4[q]; }
  }
  __syncthreads();
  auto put = [&](int s, int k, int e) {
    (VEC ? stage : out)[((s & chunk_mask) << 10) + (s >> chunk_log2)] = ((unsigned)k << 16) | (unsigned)e;
  };
  if (VEC) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (v[q].x < 0) continue;
      int len[4];
      runs(v[q], len);
      const int k[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
      const int e = (q * 1024 + tid) * 4;
      int s = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (len[u] > 0) s = atomicAdd(&cnt[k[u]], len[u]);  // a run's first element draws its slots
        put(s, k[u], e + u);
        ++s;
      }
    }
  } else {
    for (int e = tid; e < mns; e += 1024) {
      const int k = ib[e];
      put(atomicAdd(&cnt[k], 1), k, e);
    }
  }
  for (int s = mns + tid; s < (1024 << chunk_log2); s += 1024)
    (VEC ? stage : out)[((s & chunk_mask) << 10) + (s >> chunk_log2)] = kInvPad;
  if (VEC) {
    __syncthreads();
    const uint4 *s4 = reinterpret_cast<const uint4 *>(stage);
    uint4 *o4 = reinterpret_cast<uint4 *>(out);
    for (int t = tid; t < (256 << chunk_log2); t += 1024) o4[t] = s4[t];
  }
}

template <int CHUNK>
__global__ void __launch_bounds__(1024)
group_points_grad_sorted_kernel(int c, int n, int mns, int c_total, int channel0,
                                const float *__restrict__ grad_out,
                                const unsigned *__restrict__ inv, float *__restrict__ grad_points) {
  __shared__ __attribute__((aligned(16))) float row[CHUNK * 1024];
  __shared__ float acc[kInvMaxPoints];
  const BlockId blk = xcd_block_id();  // the channels of a cloud share its entries in one L2
  const int l = blk.x, b = blk.y;
  const int tid = threadIdx.x;
  const unsigned *ent = inv + (size_t)b * CHUNK * 1024 + tid;
  unsigned e[CHUNK];
#pragma clang loop unroll(full)
  for (int j = 0; j < CHUNK; ++j) e[j] = ent[j * 1024];
  // channels channel0 .. channel0+c-1 of a (b, c_total, m, ns) tensor: the feature slice of a
  // grouped tensor's gradient is read in place
  const float *g = grad_out + ((size_t)b * c_total + channel0 + l) * mns;
  if ((mns & 3) == 0) {
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    float4 *r4 = reinterpret_cast<float4 *>(row);
    for (int t = tid; t < mns / 4; t += 1024) r4[t] = g4[t];
  } else {
    for (int t = tid; t < mns; t += 1024) row[t] = g[t];
  }
  for (int t = tid; t < n; t += 1024) acc[t] = 0.f;
  __syncthreads();
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
  if (cur != (kInvPad >> 16)) atomicAdd(&acc[cur], sum);  // may continue in the next lane
  __syncthreads();
  float *dst = grad_points + ((size_t)b * c + l) * n;
  for (int t = tid; t < n; t += 1024) dst[t] = acc[t];
}

// Fused tail of QueryAndGroup (pointnet2_utils.py:348-358): given idx, write the
// (b, 3+c, m, ns) tensor: channels 0..2 = xyz[idx] - centroid (optionally * 1/radius),
// channels 3.. = features[idx].
// (written once, read by the next kernel from HBM anyway: streaming stores)
typedef float gc_f4 __attribute__((ext_vector_type(4)));

template <bool VEC>
__global__ void __launch_bounds__(256)
group_concat_kernel(int c, int n, int m, int ns, float inv_radius, int normalize, int skip_xyz,
                    const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                    const float *__restrict__ features, const int *__restrict__ idx,
                    float *__restrict__ out) {
  const BlockId blk = xcd_block_id();
  const int b = blk.z;
  const int mns = m * ns;
  const int e = (blk.x * 256 + threadIdx.x) * 4;
  if (e >= mns) return;
  const int live = mns - e < 4 ? mns - e : 4;
  const int *ib = idx + (size_t)b * mns + e;
  int ii[4] = {0, 0, 0, 0};
  if (VEC) {
    const int4 v = *reinterpret_cast<const int4 *>(ib);
    ii[0] = v.x; ii[1] = v.y; ii[2] = v.z; ii[3] = v.w;
  } else {
    for (int t = 0; t < live; ++t) ii[t] = ib[t];
  }
  const int ctot = 3 + c;
  if (blk.y == 0 && !skip_xyz) {
    const float *pts = xyz + (size_t)b * n * 3;
    const float *ctr = new_xyz + (size_t)b * m * 3;
    float r[3][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = (e + (t < live ? t : 0)) / ns;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float v = __fsub_rn(pts[ii[t] * 3 + d], ctr[j * 3 + d]);
        if (normalize) v = __fmul_rn(v, inv_radius);
        r[d][t] = v;
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float *dst = out + ((size_t)b * ctot + d) * mns + e;
      if (VEC) {
        __builtin_nontemporal_store(gc_f4{r[d][0], r[d][1], r[d][2], r[d][3]}, reinterpret_cast<gc_f4 *>(dst));
      } else {
        for (int t = 0; t < live; ++t) dst[t] = r[d][t];
      }
    }
  }
  for (int l = blk.y; l < c; l += gridDim.y) {
    const float *src = features + ((size_t)b * c + l) * n;
    float *dst = out + ((size_t)b * ctot + 3 + l) * mns + e;
    if (VEC) {
      __builtin_nontemporal_store(gc_f4{src[ii[0]], src[ii[1]], src[ii[2]], src[ii[3]]}, reinterpret_cast<gc_f4 *>(dst));
    } else {
      for (int t = 0; t < live; ++t) dst[t] = src[ii[t]];
    }
  }
}

constexpr int kGroupLdsFloats = 40960;  // 160 KB, all of a CU's LDS: one channel of N <= 40960
constexpr int kGroupLdsSmall = 16384;   // 64 KB variant: two workgroups per CU

int channel_groups(int c, int per_thread) {
  int g = (c + per_thread - 1) / per_thread;
  if (g < 1) g = 1;
  if (g > 65535) g = 65535;
  return g;
}

}  // namespace

int pn2_ball_query_grid_try(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                            const float *xyz, int *idx, void *workspace, size_t workspace_bytes,
                            hipStream_t stream, int prebuilt, int *handled);
int pn2_query_group_grid_try(int b, int n, int m, int c_gather, int ctot, float radius,
                             int nsample, int normalize_xyz, const float *new_xyz,
                             const float *xyz, const float *features, int *idx, float *out,
                             void *workspace, size_t workspace_bytes, hipStream_t stream,
                             int prebuilt, int *handled);
size_t pn2_ball_query_grid_workspace(int b, int n, int m, int nsample);
size_t pn2_grid_layout_bytes(int b, int n);
int pn2_grid_build_launch(int b, int n, float radius, const float *xyz, void *workspace,
                          hipStream_t stream);

PN2_API size_t pn2_ball_query_workspace_bytes(int b, int n, int m, int nsample) {
  return pn2_ball_query_grid_workspace(b, n, m, nsample);
}

PN2_API int pn2_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                           const float *xyz, int *idx, void *workspace, size_t workspace_bytes,
                           void *stream_) {
  if (b <= 0 || m <= 0 || nsample <= 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) {
    return pn2_zero_async(idx, sizeof(int) * (size_t)b * m * nsample, stream);
  }
  int handled = 0;
  const int rc = pn2_ball_query_grid_try(b, n, m, radius, nsample, new_xyz, xyz, idx, workspace,
                                         workspace_bytes, stream, 0, &handled);
  if (rc != 0 || handled) return rc;

  const float radius2 = radius * radius;  // fp32 product, as ball_query_gpu.cu:27
  // centroids per wave: as many as keeps >= ~1024 workgroups in flight
  const long long total = (long long)b * m;
  int qw = 8;
  while (qw > 1 && total / (4 * qw) < 1024) qw >>= 1;
  dim3 grid(pn2_ceil_div(m, 4 * qw), b);
  switch (qw) {
    case 8:
      hipLaunchKernelGGL(ball_query_bf_kernel<8>, grid, dim3(256), 0, stream, n, m, radius2,
                         nsample, new_xyz, xyz, idx);
      break;
    case 4:
      hipLaunchKernelGGL(ball_query_bf_kernel<4>, grid, dim3(256), 0, stream, n, m, radius2,
                         nsample, new_xyz, xyz, idx);
      break;
    case 2:
      hipLaunchKernelGGL(ball_query_bf_kernel<2>, grid, dim3(256), 0, stream, n, m, radius2,
                         nsample, new_xyz, xyz, idx);
      break;
    default:
      hipLaunchKernelGGL(ball_query_bf_kernel<1>, grid, dim3(256), 0, stream, n, m, radius2,
                         nsample, new_xyz, xyz, idx);
  }
  return pn2_launch_status();
}

template <int LDSF>
static void launch_group_lds(dim3 grid, int c, int n, long long mns, const float *points,
                             const int *idx, float *out, hipStream_t stream) {
  if (mns % 4 == 0)
    hipLaunchKernelGGL((group_points_lds_kernel<LDSF, true>), grid, dim3(1024), 0, stream, c, n,
                       (int)mns, points, idx, out);
  else
    hipLaunchKernelGGL((group_points_lds_kernel<LDSF, false>), grid, dim3(1024), 0, stream, c, n,
                       (int)mns, points, idx, out);
}

PN2_API int pn2_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                             const int *idx, float *out, void *stream_) {
  const long long mns = (long long)npoints * nsample;
  if (b <= 0 || c <= 0 || mns <= 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= kGroupLdsFloats && mns >= 4096 && c <= 65535) {
    // one resident round: slices so that (slices x channels x clouds) ~ the CU count, but at
    // least 1024 four-index chunks per workgroup
    const long long rows = (long long)b * c;
    long long slices = rows >= 256 ? 1 : 256 / rows;
    const long long max_slices = (mns / 4 + 1023) / 1024;
    if (slices > max_slices) slices = max_slices;
    const dim3 grid((unsigned)slices, c, b);
    if (n <= kGroupLdsSmall)
      launch_group_lds<kGroupLdsSmall>(grid, c, n, mns, points, idx, out, stream);
    else
      launch_group_lds<kGroupLdsFloats>(grid, c, n, mns, points, idx, out, stream);
    return pn2_launch_status();
  }
  dim3 grid(pn2_ceil_div(mns, 1024), channel_groups(c, 8), b);
  if (mns % 4 == 0)
    hipLaunchKernelGGL(group_points_kernel<true>, grid, dim3(256), 0, stream, c, n,
                       (int)mns, points, idx, out);
  else
    hipLaunchKernelGGL(group_points_kernel<false>, grid, dim3(256), 0, stream, c,
                       n, (int)mns, points, idx, out);
  return pn2_launch_status();
}

template <int CPW>
static void launch_grad_lds(int b, int c, int n, long long mns, const float *grad_out,
                            const int *idx, float *grad_points, hipStream_t stream) {
  dim3 grid(pn2_ceil_div(c, CPW), b);
  const size_t lds = sizeof(float) * (size_t)CPW * n;
  if (mns % 4 == 0)
    hipLaunchKernelGGL((group_points_grad_lds_kernel<CPW, true>), grid, dim3(1024), lds, stream, c,
                       n, (int)mns, grad_out, idx, grad_points);
  else
    hipLaunchKernelGGL((group_points_grad_lds_kernel<CPW, false>), grid, dim3(1024), lds, stream,
                       c, n, (int)mns, grad_out, idx, grad_points);
}

PN2_API int pn2_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                  const float *grad_out, const int *idx, float *grad_points,
                                  void *stream_) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  const long long mns = (long long)npoints * nsample;
  if (mns > 0 && n <= 16384) {
    // LDS-privatised rows: the largest channel group that fits 64 KiB and still leaves
    // >= 512 workgroups (2 per CU) in flight
    int cpw = 8;
    while (cpw > 1 && ((long long)cpw * n * 4 > 65536 || (long long)b * ((c + cpw - 1) / cpw) < 512))
      cpw >>= 1;
    switch (cpw) {
      case 8: launch_grad_lds<8>(b, c, n, mns, grad_out, idx, grad_points, stream); break;
      case 4: launch_grad_lds<4>(b, c, n, mns, grad_out, idx, grad_points, stream); break;
      case 2: launch_grad_lds<2>(b, c, n, mns, grad_out, idx, grad_points, stream); break;
      default: launch_grad_lds<1>(b, c, n, mns, grad_out, idx, grad_points, stream);
    }
    return pn2_launch_status();
  }
  if (mns > 0 && n <= kGroupLdsFloats && (long long)b * c < 128) {  // few rows: ranges of a row per workgroup
    int slices = 4;
    while ((n + slices - 1) / slices > 10240) slices *= 2;
    const dim3 grid(slices * c, b);
    if (mns % 4 == 0)
      hipLaunchKernelGGL(group_points_grad_lds_range_kernel<true>, grid, dim3(1024), 0, stream, c, n,
                         (int)mns, slices, grad_out, idx, grad_points);
    else
      hipLaunchKernelGGL(group_points_grad_lds_range_kernel<false>, grid, dim3(1024), 0, stream, c, n,
                         (int)mns, slices, grad_out, idx, grad_points);
    return pn2_launch_status();
  }
  if (mns > 0 && n <= kGroupLdsFloats && c <= 65535) {  // one row per workgroup, 160 KB of LDS
    const dim3 grid(c, b);
    if (mns % 4 == 0)
      hipLaunchKernelGGL(group_points_grad_lds_big_kernel<true>, grid, dim3(1024), 0, stream, c, n,
                         (int)mns, grad_out, idx, grad_points);
    else
      hipLaunchKernelGGL(group_points_grad_lds_big_kernel<false>, grid, dim3(1024), 0, stream, c, n,
                         (int)mns, grad_out, idx, grad_points);
    return pn2_launch_status();
  }
  const int e = pn2_zero_async(grad_points, sizeof(float) * (size_t)b * c * n, stream);
  if (e != 0) return e;
  if (mns <= 0) return 0;
  dim3 grid(pn2_ceil_div(mns, 1024), channel_groups(c, 8), b);
  if (mns % 4 == 0)
    hipLaunchKernelGGL(group_points_grad_kernel<true>, grid, dim3(256), 0, stream, c, n, (int)mns,
                       grad_out, idx, grad_points);
  else
    hipLaunchKernelGGL(group_points_grad_kernel<false>, grid, dim3(256), 0, stream, c, n,
                       (int)mns, grad_out, idx, grad_points);
  return pn2_launch_status();
}

static int group_concat_launch(int b, int n, int m, int c, float radius, int nsample,
                               int normalize_xyz, int skip_xyz, const float *new_xyz,
                               const float *xyz, const float *features, const int *idx, float *out,
                               hipStream_t stream) {
  const long long mns = (long long)m * nsample;
  dim3 grid(pn2_ceil_div(mns, 1024), channel_groups(c, 8), b);
  const float inv_radius = 1.0f / radius;  // torch divides by a scalar as x * (1/r)
  if (mns % 4 == 0)
    hipLaunchKernelGGL(group_concat_kernel<true>, grid, dim3(256), 0, stream, c, n, m, nsample,
                       inv_radius, normalize_xyz, skip_xyz, new_xyz, xyz, features, idx, out);
  else
    hipLaunchKernelGGL(group_concat_kernel<false>, grid, dim3(256), 0, stream, c, n, m, nsample,
                       inv_radius, normalize_xyz, skip_xyz, new_xyz, xyz, features, idx, out);
  return pn2_launch_status();
}

// feature channels the fused cell-list kernel gathers itself (one scattered 4-byte load per
// channel and lane); wider feature tensors go through the channel-parallel gather kernel
constexpr int kFusedGatherChannels = 8;

static int query_and_group_impl(int b, int n, int m, int c, float radius, int nsample,
                                int normalize_xyz, const float *new_xyz, const float *xyz,
                                const float *features, int *idx, float *out, void *workspace,
                                size_t workspace_bytes, int prebuilt, void *stream_) {
  if (b <= 0 || m <= 0 || nsample <= 0) return 0;
  if (c > 0 && !features) return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  if (n > 0) {  // large clouds: ball query and gathers in one kernel on the cell lists
    int handled = 0;
    const int cg = c <= kFusedGatherChannels ? c : 0;
    int rc = pn2_query_group_grid_try(b, n, m, cg, 3 + c, radius, nsample, normalize_xyz, new_xyz,
                                      xyz, features, idx, out, workspace, workspace_bytes, stream,
                                      prebuilt, &handled);
    if (rc != 0) return rc;
    if (handled) {
      if (cg == c) return 0;
      return group_concat_launch(b, n, m, c, radius, nsample, normalize_xyz, 1, new_xyz, xyz,
                                 features, idx, out, stream);
    }
  }
  if (prebuilt) return (int)hipErrorInvalidValue;  // the cell lists do not cover this shape
  int rc = pn2_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, workspace,
                          workspace_bytes, stream_);
  if (rc != 0) return rc;
  return group_concat_launch(b, n, m, c, radius, nsample, normalize_xyz, 0, new_xyz, xyz, features,
                             idx, out, stream);
}

PN2_API int pn2_query_and_group(int b, int n, int m, int c, float radius, int nsample,
                                int normalize_xyz, const float *new_xyz, const float *xyz,
                                const float *features, int *idx, float *out, void *workspace,
                                size_t workspace_bytes, void *stream_) {
  return query_and_group_impl(b, n, m, c, radius, nsample, normalize_xyz, new_xyz, xyz, features,
                              idx, out, workspace, workspace_bytes, 0, stream_);
}

// ---- cell lists as an object: built once (by pn2_grid_build or as a by-product of
// pn2_furthest_point_sampling_grid), queried by any number of ball queries of the same radius
PN2_API size_t pn2_grid_bytes(int b, int n) { return pn2_grid_layout_bytes(b, n); }

void pn2_grid_order_layout(int b, int n, size_t *start_off, size_t *order_off, int *start_stride,
                           int *order_for_slot);
PN2_API int pn2_grid_launch_order(int b, int n, size_t *start_offset, int *start_stride,
                                  int *order_for_slot, size_t *order_offset) {
  if (pn2_grid_layout_bytes(b, n) == 0 || !start_offset || !start_stride || !order_for_slot || !order_offset)
    return (int)hipErrorInvalidValue;
  pn2_grid_order_layout(b, n, start_offset, order_offset, start_stride, order_for_slot);
  return 0;
}

PN2_API int pn2_grid_build(int b, int n, float radius, const float *xyz, void *grid,
                           size_t grid_bytes, void *stream_) {
  const size_t need = pn2_grid_layout_bytes(b, n);
  if (b <= 0) return 0;
  if (need == 0 || !grid || grid_bytes < need || !(radius > 1e-6f) || !(radius < 1e6f))
    return (int)hipErrorInvalidValue;
  return pn2_grid_build_launch(b, n, radius, xyz, grid, (hipStream_t)stream_);
}

PN2_API int pn2_ball_query_prebuilt(int b, int n, int m, float radius, int nsample,
                                    const float *new_xyz, const float *xyz, int *idx,
                                    const void *grid, size_t grid_bytes, void *stream_) {
  if (b <= 0 || m <= 0 || nsample <= 0) return 0;
  int handled = 0;
  const int rc = pn2_ball_query_grid_try(b, n, m, radius, nsample, new_xyz, xyz, idx,
                                         const_cast<void *>(grid), grid_bytes,
                                         (hipStream_t)stream_, 1, &handled);
  if (rc != 0) return rc;
  return handled ? 0 : (int)hipErrorInvalidValue;
}

PN2_API int pn2_query_and_group_prebuilt(int b, int n, int m, int c, float radius, int nsample,
                                         int normalize_xyz, const float *new_xyz,
                                         const float *xyz, const float *features, int *idx,
                                         float *out, const void *grid, size_t grid_bytes,
                                         void *stream_) {
  return query_and_group_impl(b, n, m, c, radius, nsample, normalize_xyz, new_xyz, xyz, features,
                              idx, out, const_cast<void *>(grid), grid_bytes, 1, stream_);
}

PN2_API int pn2_query_and_group_picks(int b, int n, int m, int c, float radius, int nsample,
                                      int normalize_xyz, const float *new_xyz, const float *xyz,
                                      const float *features, int *idx, float *out,
                                      const void *grid, size_t grid_bytes, void *stream_) {
  return query_and_group_impl(b, n, m, c, radius, nsample, normalize_xyz, new_xyz, xyz, features,
                              idx, out, const_cast<void *>(grid), grid_bytes, 2, stream_);
}

// ---- inverse index of an index array (group_points_gpu.cu:48-69 does a same-address atomic per
// element instead) ------------------------------------------------------------------------------
PN2_API int pn2_group_inverse_supported(int n, int npoints, int nsample) {
  const long long mns = (long long)npoints * nsample;
  return n > 0 && n <= kInvMaxPoints && mns > 0 && mns <= kInvMaxEntries;
}

PN2_API int pn2_group_inverse_entries(int npoints, int nsample) {
  const long long mns = (long long)npoints * nsample;
  if (mns <= 0 || mns > kInvMaxEntries) return 0;
  return inverse_chunk((int)mns) * 1024;
}

PN2_API int pn2_group_inverse_build(int b, int n, int npoints, int nsample, const int *idx,
                                    unsigned *inv, void *stream_) {
  if (b <= 0) return 0;
  if (!pn2_group_inverse_supported(n, npoints, nsample)) return (int)hipErrorInvalidValue;
  const int chunk = inverse_chunk(npoints * nsample);
  const int chunk_log2 = chunk == 4 ? 2 : chunk == 8 ? 3 : chunk == 16 ? 4 : 5;
  if ((npoints * nsample) % 4 == 0)
    hipLaunchKernelGGL(group_inverse_kernel<true>, dim3(b), dim3(1024), 0, (hipStream_t)stream_, n,
                       npoints * nsample, chunk_log2, idx, inv);
  else
    hipLaunchKernelGGL(group_inverse_kernel<false>, dim3(b), dim3(1024), 0, (hipStream_t)stream_, n,
                       npoints * nsample, chunk_log2, idx, inv);
  return pn2_launch_status();
}

PN2_API int pn2_group_points_grad_sorted(int b, int c, int n, int npoints, int nsample,
                                         const float *grad_out, int c_total, int channel0,
                                         const unsigned *inv, float *grad_points, void *stream_) {
  if (b <= 0 || c <= 0) return 0;
  if (!pn2_group_inverse_supported(n, npoints, nsample) || c > 65535 || channel0 < 0 ||
      channel0 + c > c_total)
    return (int)hipErrorInvalidValue;
  const int mns = npoints * nsample;
  const dim3 grid(c, b);
  hipStream_t stream = (hipStream_t)stream_;
  switch (inverse_chunk(mns)) {
    case 4:
      hipLaunchKernelGGL(group_points_grad_sorted_kernel<4>, grid, dim3(1024), 0, stream, c, n,
                         mns, c_total, channel0, grad_out, inv, grad_points);
      break;
    case 8:
      hipLaunchKernelGGL(group_points_grad_sorted_kernel<8>, grid, dim3(1024), 0, stream, c, n,
                         mns, c_total, channel0, grad_out, inv, grad_points);
      break;
    case 16:
      hipLaunchKernelGGL(group_points_grad_sorted_kernel<16>, grid, dim3(1024), 0, stream, c, n,
                         mns, c_total, channel0, grad_out, inv, grad_points);
      break;
    default:
      hipLaunchKernelGGL(group_points_grad_sorted_kernel<32>, grid, dim3(1024), 0, stream, c, n,
                         mns, c_total, channel0, grad_out, inv, grad_points);
  }
  return pn2_launch_status();
}

PN2_API int pn2_group_concat(int b, int n, int m, int c, float radius, int nsample,
                             int normalize_xyz, const float *new_xyz, const float *xyz,
                             const float *features, const int *idx, float *out, void *stream_) {
  if (b <= 0 || m <= 0 || nsample <= 0) return 0;
  if (c > 0 && !features) return (int)hipErrorInvalidValue;
  return group_concat_launch(b, n, m, c, radius, nsample, normalize_xyz, 0, new_xyz, xyz, features,
                             idx, out, (hipStream_t)stream_);
}
