// tools/micro/lds_read_rate.hip -- LDS read throughput per CU by instruction: ds_read_b64_tr_b16 (the
// transposing fragment read of the bf16 images), ds_read_b64, ds_read_b128; conflict-free addresses,
// 8 waves per workgroup, one workgroup per CU.  bytes = workgroups * 512 * iters * READS * width.
#include <hip/hip_runtime.h>
typedef short b4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(512) lds_loop(int iters, float *out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int t = threadIdx.x; t < 16384; t += 512) reinterpret_cast<unsigned *>(lds)[t] = t;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0;
  // tr: the image layout of the kernels (row pitch 64 bytes): 8 rows x 64 bytes per read
  const int tr_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  for (int it = 0; it < iters; ++it) {
    const int base = ((it + wave) & 7) * 8192;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (KIND == 0) {
        const b4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) b4v *)(__attribute__((address_space(3))) char *)(lds + base + r * 1024 + tr_off));
        acc += (unsigned)v[0] ^ (unsigned)v[3];
      } else if (KIND == 1) {
        const uint2 v = *reinterpret_cast<const uint2 *>(lds + base + r * 512 + lane * 8);
        acc += v.x ^ v.y;
      } else {
        const u4v v = *reinterpret_cast<const u4v *>(lds + base + (r & 7) * 1024 + lane * 16);
        acc += v[0] ^ v[3];
      }
    }
  }
  if (acc == 0x12345678u) out[threadIdx.x] = 1.f;
}

extern "C" int lds_read_rate_launch(int kind, int blocks, int iters, float *out, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = 65536;
  if (kind == 0) hipLaunchKernelGGL((lds_loop<0>), dim3(blocks), dim3(512), bytes, st, iters, out);
  else if (kind == 1) hipLaunchKernelGGL((lds_loop<1>), dim3(blocks), dim3(512), bytes, st, iters, out);
  else hipLaunchKernelGGL((lds_loop<2>), dim3(blocks), dim3(512), bytes, st, iters, out);
  return (int)hipGetLastError();
}
