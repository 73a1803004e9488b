// tools/micro/td_rate.hip -- how fast does a CU's vector-memory return path deliver 64-lane row
// loads of 16 / 12 / 8 / 4 bytes per lane from L2-resident data?  (design question behind the
// pair kernel: would 8-byte candidate records halve its time?)  One wave per "centroid", nine row
// loads at pseudo-random 16-byte-aligned offsets inside a 640 KB array per cloud, results reduced
// and written as one dword per wave so that nothing is optimised away.
#include <hip/hip_runtime.h>
#include <stdint.h>

// STORES: 0 = one dword per wave; 5 = five 64-lane dword stores (256 B each, five planes);
//         2 = one 64-lane dwordx4 store (four planes, 16 lanes x 16 B each) + one dword store
template <int BYTES, int STORES = 0, int WPB = 1>
__global__ void __launch_bounds__(64 * WPB)
td_rate_kernel(const char *__restrict__ base, unsigned cloud_bytes, int m, int rows,
               float *__restrict__ out) {
  const int wg = blockIdx.x * WPB + (threadIdx.x >> 6), b = wg / m, j = wg - b * m;
  const int lane = threadIdx.x & 63;
  const char *cloud = base + (size_t)b * cloud_bytes;
  unsigned h = (unsigned)j * 2654435761u + 12345u;
  float acc = 0.f;
  for (int r = 0; r < rows; ++r) {
    h = h * 1664525u + 1013904223u;
    const unsigned off = ((h >> 8) % (cloud_bytes / 16 - 64)) * 16u;  // wave-uniform row start
    const char *p = cloud + off + (unsigned)lane * BYTES;
    if (BYTES == 16) { const float4 v = *reinterpret_cast<const float4 *>(p); acc += v.x + v.y + v.z + v.w; }
    else if (BYTES == 12) { const float *q = reinterpret_cast<const float *>(p); struct __attribute__((packed, aligned(4))) f3 { float x, y, z; }; const f3 v = *reinterpret_cast<const f3 *>(q); acc += v.x + v.y + v.z; }
    else if (BYTES == 8) { const float2 v = *reinterpret_cast<const float2 *>(p); acc += v.x + v.y; }
    else { acc += *reinterpret_cast<const float *>(p); }
  }
  if (STORES == 5) {
    const size_t plane = (size_t)gridDim.x * WPB * 64;
    for (int p = 0; p < 5; ++p) out[p * plane + (size_t)wg * 64 + lane] = acc + p;
    return;
  }
  if (STORES == 2) {
    const size_t plane = (size_t)gridDim.x * 64;
    const int p = lane >> 4, q = lane & 15;
    *reinterpret_cast<float4 *>(out + p * plane + (size_t)wg * 64 + q * 4) = make_float4(acc, acc + 1, acc + 2, acc + 3);
    out[4 * plane + (size_t)wg * 64 + lane] = acc;
    return;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) out[wg] = acc;
}

// the same nine 16-byte row loads as LDS-direct loads (global_load_lds_dwordx4: the data goes
// from the memory pipeline into LDS without passing through VGPRs), read back from LDS
template <int STORES>
__global__ void __launch_bounds__(64)
td_rate_lds_kernel(const char *__restrict__ base, unsigned cloud_bytes, int m, int rows,
                   float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float4 stage[9 * 64];
  const int wg = blockIdx.x, b = wg / m, j = wg - b * m;
  const int lane = threadIdx.x & 63;
  const char *cloud = base + (size_t)b * cloud_bytes;
  unsigned h = (unsigned)j * 2654435761u + 12345u;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    h = h * 1664525u + 1013904223u;
    const unsigned off = ((h >> 8) % (cloud_bytes / 16 - 64)) * 16u;
    const char *p = cloud + off + (unsigned)lane * 16;
    if (r < rows)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                       (__attribute__((address_space(3))) void *)(stage + r * 64), 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 9; ++r)
    if (r < rows) { const float4 v = stage[r * 64 + lane]; acc += v.x + v.y + v.z + v.w; }
  if (STORES == 5) {
    const size_t plane = (size_t)gridDim.x * 64;
    for (int p = 0; p < 5; ++p) out[p * plane + (size_t)wg * 64 + lane] = acc + p;
    return;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) out[wg] = acc;
}

extern "C" __attribute__((visibility("default")))
int td_rate_launch(int bytes, const void *base, unsigned cloud_bytes, int b, int m, int rows,
                   float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const char *p = (const char *)base;
  if (bytes == 16) hipLaunchKernelGGL(td_rate_kernel<16>, dim3(b * m), dim3(64), 0, s, p, cloud_bytes, m, rows, out);
  else if (bytes == 12) hipLaunchKernelGGL(td_rate_kernel<12>, dim3(b * m), dim3(64), 0, s, p, cloud_bytes, m, rows, out);
  else if (bytes == 8) hipLaunchKernelGGL(td_rate_kernel<8>, dim3(b * m), dim3(64), 0, s, p, cloud_bytes, m, rows, out);
  else if (bytes == 4) hipLaunchKernelGGL(td_rate_kernel<4>, dim3(b * m), dim3(64), 0, s, p, cloud_bytes, m, rows, out);
  else if (bytes == 165) hipLaunchKernelGGL((td_rate_kernel<16, 5>), dim3(b * m), dim3(64), 0, s, p, cloud_bytes, m, rows, out);
  else if (bytes == 162) hipLaunchKernelGGL((td_rate_kernel<16, 2>), dim3(b * m), dim3(64), 0, s, p, cloud_bytes, m, rows, out);
  else if (bytes == 1654) hipLaunchKernelGGL((td_rate_kernel<16, 5, 4>), dim3(b * m / 4), dim3(256), 0, s, p, cloud_bytes, m, rows, out);
  else if (bytes == 16516) hipLaunchKernelGGL((td_rate_kernel<16, 5, 16>), dim3(b * m / 16), dim3(1024), 0, s, p, cloud_bytes, m, rows, out);
  else if (bytes == 1600) hipLaunchKernelGGL((td_rate_lds_kernel<0>), dim3(b * m), dim3(64), 0, s, p, cloud_bytes, m, rows, out);
  else if (bytes == 1605) hipLaunchKernelGGL((td_rate_lds_kernel<5>), dim3(b * m), dim3(64), 0, s, p, cloud_bytes, m, rows, out);
  return (int)hipGetLastError();
}
