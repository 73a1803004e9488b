#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short *out, const int *addr) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)((__attribute__((address_space(3))) char *)lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main(int argc, char **argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;
  int h[64]; 
  for (int l = 0; l < 64; ++l) {
    if (mode == 0) h[l] = 0;                  // uniform
    else if (mode == 1) h[l] = l * 8;         // consecutive 8-byte units
    else if (mode == 2) h[l] = (l & 15) * 64 + (l >> 4) * 8;  // row l&15 (stride 64 B), 8-byte column block l>>4
    else h[l] = l * 32;
  }
  int *d; unsigned short *o; hipMalloc(&d, 256); hipMalloc(&o, 512);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(o, d);
  unsigned short r[256]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h[l], r[4*l], r[4*l+1], r[4*l+2], r[4*l+3]);
}
