// tools/micro/mfma_bf16_peak.hip -- what v_mfma_f32_32x32x16_bf16 sustains on this box: waves that do
// nothing but issue MFMAs, ACC independent accumulators per wave (1 = one dependent chain), optionally
// VALU instructions between them (VPER vector instructions behind every MFMA, independent of it).
// FLOPs = waves * iters * ACC * 32768.
#include <hip/hip_runtime.h>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef short b8v __attribute__((ext_vector_type(8)));

template <int ACC, int VPER>
__global__ void __launch_bounds__(256) mfma_loop(int iters, float *out, const float *in) {
  f16v acc[ACC];
#pragma unroll
  for (int a = 0; a < ACC; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
  b8v av, bv;
#pragma unroll
  for (int e = 0; e < 8; ++e) { av[e] = (short)(threadIdx.x + e); bv[e] = (short)(threadIdx.x * 3 + e); }
  float x0 = in[threadIdx.x], x1 = in[threadIdx.x + 256], x2 = 1.f, x3 = 2.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < ACC; ++a) {
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[a], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < VPER; ++v) {
        // four independent dependent-chains of plain fp32 instructions
        if ((v & 3) == 0) x0 = __builtin_fmaf(x0, 1.0001f, 0.5f);
        if ((v & 3) == 1) x1 = __builtin_fmaf(x1, 0.9999f, 0.25f);
        if ((v & 3) == 2) x2 = __builtin_fmaf(x2, 1.0002f, 0.125f);
        if ((v & 3) == 3) x3 = __builtin_fmaf(x3, 0.9998f, 0.75f);
      }
      if (VPER) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = x0 + x1 + x2 + x3;
#pragma unroll
  for (int a = 0; a < ACC; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[a][q];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// the x6 pattern: six consecutive MFMAs into one accumulator, then the next accumulator (CHAIN = 6),
// against the same twelve MFMAs alternating between two accumulators (CHAIN = 1)
template <int CHAIN>
__global__ void __launch_bounds__(256) mfma_x6_pattern(int iters, float *out, const float *in) {
  f16v acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
  b8v av, bv;
#pragma unroll
  for (int e = 0; e < 8; ++e) { av[e] = (short)(threadIdx.x + e); bv[e] = (short)(threadIdx.x * 3 + e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      if (CHAIN == 6) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int t = 0; t < 6; ++t) acc[2 * pr + a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[2 * pr + a], 0, 0, 0);
      } else {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int a = 0; a < 2; ++a) acc[2 * pr + a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[2 * pr + a], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[a][q];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

extern "C" int mfma_bf16_peak_launch(int variant, int blocks, int iters, float *out, const float *in, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
#define V(N, A, P) case N: hipLaunchKernelGGL((mfma_loop<A, P>), dim3(blocks), dim3(256), 0, st, iters, out, in); break
    V(0, 1, 0); V(1, 2, 0); V(2, 4, 0); V(3, 1, 2); V(4, 1, 4); V(5, 1, 6); V(6, 1, 8); V(7, 2, 4); V(8, 2, 6); V(9, 4, 6);
#undef V
    case 10: hipLaunchKernelGGL((mfma_x6_pattern<6>), dim3(blocks), dim3(256), 0, st, iters, out, in); break;
    case 11: hipLaunchKernelGGL((mfma_x6_pattern<1>), dim3(blocks), dim3(256), 0, st, iters, out, in); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
