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
