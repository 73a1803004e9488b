// tools/micro/mfma_war.hip -- does a vector instruction that OVERWRITES a source register of an MFMA
// issued just before it change that MFMA's result on gfx950?  (The compiler's hazard tables have no
// write-after-read rule for SrcA / SrcB; hand-interleaved kernels put such writes one instruction
// behind the MFMA.)  NQ independent MFMAs are issued back to back (so that the last ones queue behind
// the matrix pipe), then the B operand of the LAST is overwritten DIST instructions later.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NQ, int DIST>
__global__ void __launch_bounds__(64) war_kernel(const int *in, float *out, int poison) {
  const int lane = threadIdx.x;
  bf16x8 a, b[NQ];
  for (int j = 0; j < 8; ++j) a[j] = (short)in[lane * 8 + j];
  for (int q = 0; q < NQ; ++q)
    for (int j = 0; j < 8; ++j) b[q][j] = (short)in[512 + ((lane * 8 + j + 37 * q) & 511)];
  f32x16 acc[NQ];
  for (int q = 0; q < NQ; ++q)
    for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 bl = __builtin_bit_cast(i32x4, b[NQ - 1]);
  const int pz = poison;
  // the last MFMA's B operand sits in v[20:23] (fixed: the write below must name one of its registers)
  asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\ts_nop 7"
               :: "v"(bl[0]), "v"(bl[1]), "v"(bl[2]), "v"(bl[3]) : "v20", "v21", "v22", "v23");
  // NQ - 1 independent MFMAs back to back, so that the last one queues behind the matrix pipe
#pragma unroll
  for (int q = 0; q < NQ - 1; ++q)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[q]) : "v"(a), "v"(b[q]) : "v20", "v21", "v22", "v23");
  if (DIST == 1)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, v[20:23], %0\n\t"
                 "v_mov_b32 v20, %2\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %2\n\t"
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                 : "+a"(acc[NQ - 1]) : "v"(a), "v"(pz) : "v20", "v21", "v22", "v23");
  else if (DIST == 2)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, v[20:23], %0\n\t"
                 "s_nop 0\n\t"
                 "v_mov_b32 v20, %2\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %2\n\t"
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                 : "+a"(acc[NQ - 1]) : "v"(a), "v"(pz) : "v20", "v21", "v22", "v23");
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, v[20:23], %0\n\t"
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                 "v_mov_b32 v20, %2\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %2\n\t"
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                 : "+a"(acc[NQ - 1]) : "v"(a), "v"(pz) : "v20", "v21", "v22", "v23");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[NQ - 1][i] * (float)(i + 1);
  out[blockIdx.x * 64 + lane] = s;
}

extern "C" __attribute__((visibility("default")))
int mfma_war_launch(int nq, int dist, const int *in, float *out, int poison, void *stream) {
#define L(NQ, D) hipLaunchKernelGGL((war_kernel<NQ, D>), dim3(256), dim3(64), 0, (hipStream_t)stream, in, out, poison)
  if (nq == 1 && dist == 1) L(1, 1); else if (nq == 1 && dist == 2) L(1, 2); else if (nq == 1) L(1, 3);
  else if (nq == 2 && dist == 1) L(2, 1); else if (nq == 2 && dist == 2) L(2, 2); else if (nq == 2) L(2, 3);
  else if (nq == 4 && dist == 1) L(4, 1); else if (nq == 4 && dist == 2) L(4, 2); else if (nq == 4) L(4, 3);
  else if (nq == 8 && dist == 1) L(8, 1); else if (nq == 8 && dist == 2) L(8, 2); else if (nq == 8) L(8, 3);
  else return -1;
  return (int)hipGetLastError();
}
