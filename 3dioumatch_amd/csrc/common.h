// 3dioumatch_amd/csrc/common.h -- shared helpers for the gfx950 kernels.
//
// Arithmetic contract: fp32, source order, every operation rounded.
// All translation units are built with -ffp-contract=off AND the distance / geometry
// helpers below use the explicitly rounded intrinsics, so no a*b+c is ever fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PN2_API extern "C" __attribute__((visibility("default")))

constexpr int kWave = 64;  // CDNA wavefront width

static inline int pn2_launch_status() { return (int)hipGetLastError(); }

static inline int pn2_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// XCD-aware block ids.  The dispatcher hands workgroup i (x fastest, then y, z) to XCD i % 8,
// and every XCD has a private 4 MB L2.  With the cloud index in the slowest grid dimension the
// default placement makes every XCD touch every cloud (each L2 re-fetches all of them); this
// remap gives each XCD a CONTIGUOUS range of logical block ids instead -- for B = 8 clouds
// exactly one cloud per XCD -- so a cloud's points / cell lists are fetched from HBM once and
// then hit in that XCD's L2.  Bijective for any grid size (cdna_hip_programming.md T1); purely
// a placement choice, no kernel depends on it for correctness.
struct BlockId { int x, y, z; };

__device__ __forceinline__ int xcd_contiguous_block(int hw, int nwg) {
  const int xcd = hw & 7, slot = hw >> 3, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

__device__ __forceinline__ BlockId xcd_block_id() {
  const int gx = gridDim.x, gy = gridDim.y;
  const int nwg = gx * gy * gridDim.z;
  int l = xcd_contiguous_block(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), nwg);
  BlockId o;
  o.x = l % gx;
  l /= gx;
  o.y = l % gy;
  o.z = l / gy;
  return o;
}

// Zero `bytes` (a multiple of 4) of device memory with an ordinary kernel.  Used instead of
// hipMemsetAsync so that, when the caller's stream is being captured into a HIP graph, the
// clear is a plain kernel node ordered like every other launch.
static __global__ void __launch_bounds__(256) pn2_zero_words_kernel(unsigned int *p, size_t words) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < words; i += stride) {
    if (i + 4 <= words && ((size_t)(p + i) & 15) == 0) {
      *reinterpret_cast<uint4 *>(p + i) = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (size_t j = i; j < words && j < i + 4; ++j) p[j] = 0u;
    }
  }
}

static inline int pn2_zero_async(void *ptr, size_t bytes, hipStream_t stream) {
  const size_t words = bytes / 4;
  if (words == 0) return 0;
  long long blocks = (long long)((words + 1023) / 1024);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pn2_zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     (unsigned int *)ptr, words);
  return (int)hipGetLastError();
}

// (ax-bx)^2 + (ay-by)^2 + (az-bz)^2, left to right, each op rounded to fp32.
// This is the expression every reference kernel uses for a squared distance
// (ball_query_gpu.cu:36-37, interpolate_gpu.cu:39, sampling_gpu.cu:108-109).
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by,
                                         float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mask_rank(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
