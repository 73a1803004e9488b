// tools/micro/lat_probe.hip -- clocks (s_memtime) per instruction / construct on gfx950, measured
// inside one workgroup of W waves: the cost table the FPS round loop is designed against.
// Every test times REP back-to-back copies of a construct and reports (t1 - t0) / REP for wave 0.
#include <hip/hip_runtime.h>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

extern "C" __global__ void __launch_bounds__(1024)
lat_probe_kernel(float *out, const float *in, int *chase, int zero) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  float a = in[threadIdx.x], b = in[threadIdx.x + 64], c = in[lane + 128], d = in[lane + 192];
  float e = a + 1.f, f = b + 1.f, g = c + 1.f, h = d + 1.f;
  unsigned long long t0, t1;
  int k = 0;
#define BEGIN __syncthreads(); t0 = now();
#define END(NAME_IDX, N) t1 = now(); if (threadIdx.x == 0) out[NAME_IDX] = (float)(t1 - t0) / (N); asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));

  // 0: empty (stamp cost)
  BEGIN END(0, 1)
  // 1: dependent v_add_f32
  BEGIN REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) END(1, 64)
  // 2: 4 independent v_add_f32 streams
  BEGIN REP64(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) END(2, 256)
  // 3: dependent v_max_f32_dpp (with the s_nop 1 the hazard needs)
  BEGIN REP64(asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));) END(3, 64)
  // 4: 4 independent v_max_f32_dpp streams (no nops needed between different registers)
  BEGIN REP64(asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(c), "+v"(d), "+v"(e));) END(4, 256)
  // 5: dependent v_permlane32_swap
  BEGIN REP64(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));) END(5, 64)
  // 6: dependent v_permlane16_swap
  BEGIN REP64(asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));) END(6, 64)
  // 7: v_readlane_b32 with a constant lane (independent)
  { int s0; BEGIN REP64(asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(a));) END(7, 64) k += s0; }
  // 8: VALU -> SGPR -> VALU round trip: v_readfirstlane, s_add, v_add with the SGPR
  { int s0; BEGIN REP64(asm volatile("v_readfirstlane_b32 %1, %0\n s_add_i32 %1, %1, 1\n v_add_u32 %0, %1, %0" : "+v"(k), "=s"(s0));) END(8, 64) }
  // 9: v_cmp -> vcc -> s_ff1 -> v_readlane by SGPR lane (the argmax tail)
  { int s0, s1; BEGIN REP64(asm volatile("v_cmp_ge_f32 vcc, %2, %2\n s_ff1_i32_b64 %0, vcc\n s_nop 3\n v_readlane_b32 %1, %2, %0\n v_add_f32 %2, %2, %1" : "=s"(s0), "=s"(s1), "+v"(a) :: "vcc");) END(9, 64) k += s0; }
  // 10: taken branch
  BEGIN REP64(asm volatile("s_branch 1f\n s_nop 0\n 1:");) END(10, 64)
  // 11: not-taken conditional branch
  BEGIN REP64(asm volatile("s_cmp_eq_u32 %0, 77\n s_cbranch_scc1 1f\n 1:" :: "s"(zero));) END(11, 64)
  // 12: dependent SALU chain (s_ff1 + s_bitset0)
  { unsigned long long m = ~0ull; int s0 = 0; BEGIN REP64(asm volatile("s_ff1_i32_b64 %1, %0\n s_bitset0_b64 %0, %1" : "+s"(m), "+s"(s0));) END(12, 64) k += s0; }
  // 13: LDS write -> read round trip (dependent through memory)
  BEGIN REP64(asm volatile("ds_write_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"((lane + 64 * w) * 4));) END(13, 64)
  // 14: s_barrier alone
  BEGIN REP64(asm volatile("s_barrier");) END(14, 64)
  // 15: global load latency, dependent chain through L2 (chase[] holds a cycle of offsets, 4 KiB apart)
  { int p = lane * 0 + zero; BEGIN for (int i = 0; i < 64; ++i) { p = chase[p]; } END(15, 64) k += p; }
  // 16: global load latency, sc1 (bypass L1?) not portable -- same chain a second time (L1 warm)
  { int p = zero; BEGIN for (int i = 0; i < 64; ++i) { p = chase[p]; } END(16, 64) k += p; }
  // 17: v_cndmask dependent
  BEGIN REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) END(17, 64)
  // 18: v_mul_f32 + v_add_f32 dependent pair (distance arithmetic)
  BEGIN REP64(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) END(18, 128)
  // 19: ballot -> scalar test -> uniform branch not taken (the tie check)
  BEGIN REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %0\n s_cmp_lg_u64 vcc, 0\n s_cbranch_scc1 1f\n 1:" :: "v"(a) : "vcc", "scc");) END(19, 64)
  // 20: global store (fire and forget) issue cost
  BEGIN REP64(asm volatile("global_store_dword %0, %1, off" :: "v"((unsigned long long)(out + 64 + threadIdx.x)), "v"(a) : "memory");) END(20, 64)
  // 21: s_memtime back to back
  { unsigned long long x0; BEGIN REP64(asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(x0));) END(21, 64) k += (int)x0; }
  // 22: v_mov + s_set_gpr_idx_on/off (dynamic register index)
  BEGIN REP64(asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n v_mov_b32 %0, %0\n s_set_gpr_idx_off" : "+v"(a) : "s"(zero) : "m0");) END(22, 64)
  // 23: LDS broadcast read latency only
  BEGIN REP64(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %1, 0xffc, %0" : "=v"(a), "+v"(k));) END(23, 64)

  if (k == 123456789) out[100] = a + b + c + d + e + f + g + h;
}

extern "C" int lat_probe_launch(int waves, float *out, const float *in, int *chase, void *stream) {
  hipLaunchKernelGGL(lat_probe_kernel, dim3(1), dim3(waves * 64), 0, (hipStream_t)stream, out, in, chase, 0);
  return (int)hipGetLastError();
}
