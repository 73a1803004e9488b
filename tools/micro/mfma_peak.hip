// tools/micro/mfma_peak.hip -- what v_mfma_f32_32x32x2_f32 sustains on this box (clocks included):
// waves that do nothing but issue independent MFMAs, optionally fed by LDS fragment reads the way
// the shared-MLP GEMM feeds them.  FLOPs = waves * iters * ACC * 4096.
#include <hip/hip_runtime.h>

typedef float f16v __attribute__((ext_vector_type(16)));

template <int ACC, int LDSREADS>
__global__ void __launch_bounds__(256) mfma_loop(int iters, float *out, const float *in) {
  __shared__ float frag[4096];
  for (int t = threadIdx.x; t < 4096; t += 256) frag[t] = in[t];
  __syncthreads();
  f16v acc[ACC];
#pragma unroll
  for (int a = 0; a < ACC; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
  float av = in[threadIdx.x], bv = in[threadIdx.x + 256];
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    if (LDSREADS) {
#pragma unroll
      for (int r = 0; r < LDSREADS; ++r) {
        const float v = frag[(lane + 64 * ((it + r) & 31)) & 4095];
        if (r & 1) bv = v; else av = v;
        asm volatile("" : "+v"(av), "+v"(bv));
      }
    }
#pragma unroll
    for (int a = 0; a < ACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < ACC; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[a][q];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

extern "C" int mfma_peak_launch(int variant, int blocks, int iters, float *out, const float *in, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 0: hipLaunchKernelGGL((mfma_loop<4, 0>), dim3(blocks), dim3(256), 0, st, iters, out, in); break;
    case 1: hipLaunchKernelGGL((mfma_loop<4, 2>), dim3(blocks), dim3(256), 0, st, iters, out, in); break;
    case 2: hipLaunchKernelGGL((mfma_loop<4, 4>), dim3(blocks), dim3(256), 0, st, iters, out, in); break;
    case 3: hipLaunchKernelGGL((mfma_loop<8, 0>), dim3(blocks), dim3(256), 0, st, iters, out, in); break;
    case 4: hipLaunchKernelGGL((mfma_loop<2, 0>), dim3(blocks), dim3(256), 0, st, iters, out, in); break;
    case 5: hipLaunchKernelGGL((mfma_loop<1, 0>), dim3(blocks), dim3(256), 0, st, iters, out, in); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
