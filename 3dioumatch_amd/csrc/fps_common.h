// 3dioumatch_amd/csrc/fps_common.h -- pieces shared by the furthest-point-sampling kernels
// (pn2_sampling.hip: register-resident / streaming tiers; pn2_fps_bucket.hip: bucketed tier).
//
// Exact index parity with the reference (sampling_gpu.cu:75-178): the
// reference's result depends on its reduction tree -- among equal maxima the winner minimises
// bitreverse(k mod bs) and then k, where bs = 2^floor(log2 n) capped at 512
// (cuda_utils.h:20-24).  That order is reproduced by comparing (value, key(k)) with
// key = bitrev(k mod bs) << 22 | k, independent of how lanes / waves / buckets are arranged.
#pragma once
#include "common.h"

namespace fps {

__device__ __forceinline__ unsigned fps_key(int k, int log2bs) {
  const unsigned low = (unsigned)k & ((1u << log2bs) - 1u);
  const unsigned rev = log2bs ? (__brev(low) >> (32 - log2bs)) : 0u;
  return (rev << 22) | (unsigned)k;  // k < 2^22 (checked on the host)
}

// ---- cross-lane reductions without LDS round trips -------------------------------------
// quad_perm / row_half_mirror / row_mirror DPP moves give the xor-1/2 and mirror-4/8
// exchanges inside a 16-lane row; gfx950's v_permlane16_swap / v_permlane32_swap exchange
// rows and wave halves.  Every lane ends up with the reduction of all 64 lanes.  (A shuffle
// based reduction costs 12 ds_bpermute round trips per round here; this costs none.)
template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRowHalfMirror = 0x141, kRowMirror = 0x140;

// NB: read the two results of a permlane swap into scalars before reinterpreting them;
// __builtin_bit_cast applied directly to an element of the returned vector folds both
// elements into element 0 (clang 22 / ROCm 7.2), silently dropping half of the exchange.
template <bool HALF>
__device__ __forceinline__ void swap_rows(unsigned a, unsigned b, unsigned &r0, unsigned &r1) {
  // HALF: r0 = (a.lo32, b.lo32), r1 = (a.hi32, b.hi32);  else: r0 takes b's even rows into its odd
  // rows: r0 = (a.row0, b.row0, a.row2, b.row2), r1 = (a.row1, b.row1, a.row3, b.row3)
  if (HALF) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    r0 = r[0]; r1 = r[1];
  } else {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    r0 = r[0]; r1 = r[1];
  }
}
template <bool HALF>
__device__ __forceinline__ void swap_rows(unsigned v, unsigned &r0, unsigned &r1) {
  swap_rows<HALF>(v, v, r0, r1);
}

// One reduction step = ONE instruction: v_max_f32 / v_min_f32 with the DPP-permuted operand
// (the compiler's own form is v_mov_dpp + v_cmp + v_cndmask: three dependent instructions of
// ~8 clocks each, and FPS rounds are nothing but dependent chains).  s_nop 1 covers the
// VALU-write -> DPP-read hazard, which the compiler cannot see inside an asm statement.
#define FPS_DPP_OP(OP, V, CTRL) \
  asm("s_nop 1\n\t" OP " %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(V) : "v"(V))

// max over the first LANES lanes' groups: LANES = 8 -> every lane gets the max of its aligned
// group of 8, 16 -> of its row, 64 -> of the wave
template <int LANES = 64>
__device__ __forceinline__ float wave_max_f32(float v) {
  FPS_DPP_OP("v_max_f32_dpp", v, "quad_perm:[1,0,3,2]");
  FPS_DPP_OP("v_max_f32_dpp", v, "quad_perm:[2,3,0,1]");
  FPS_DPP_OP("v_max_f32_dpp", v, "row_half_mirror");
  if (LANES > 8) FPS_DPP_OP("v_max_f32_dpp", v, "row_mirror");
  if (LANES > 16) {
    unsigned r0, r1;
    swap_rows<false>(__builtin_bit_cast(unsigned, v), r0, r1);
    float f0 = __builtin_bit_cast(float, r0), f1 = __builtin_bit_cast(float, r1);
    asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(f0), "v"(f1));
    swap_rows<true>(__builtin_bit_cast(unsigned, v), r0, r1);
    f0 = __builtin_bit_cast(float, r0); f1 = __builtin_bit_cast(float, r1);
    asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(f0), "v"(f1));
  }
  return v;
}

__device__ __forceinline__ float wave_min_f32(float v) {
  FPS_DPP_OP("v_min_f32_dpp", v, "quad_perm:[1,0,3,2]");
  FPS_DPP_OP("v_min_f32_dpp", v, "quad_perm:[2,3,0,1]");
  FPS_DPP_OP("v_min_f32_dpp", v, "row_half_mirror");
  FPS_DPP_OP("v_min_f32_dpp", v, "row_mirror");
  unsigned r0, r1;
  swap_rows<false>(__builtin_bit_cast(unsigned, v), r0, r1);
  float f0 = __builtin_bit_cast(float, r0), f1 = __builtin_bit_cast(float, r1);
  asm("v_min_f32 %0, %1, %2" : "=v"(v) : "v"(f0), "v"(f1));
  swap_rows<true>(__builtin_bit_cast(unsigned, v), r0, r1);
  f0 = __builtin_bit_cast(float, r0); f1 = __builtin_bit_cast(float, r1);
  asm("v_min_f32 %0, %1, %2" : "=v"(v) : "v"(f0), "v"(f1));
  return v;
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  FPS_DPP_OP("v_min_u32_dpp", v, "quad_perm:[1,0,3,2]");
  FPS_DPP_OP("v_min_u32_dpp", v, "quad_perm:[2,3,0,1]");
  FPS_DPP_OP("v_min_u32_dpp", v, "row_half_mirror");
  FPS_DPP_OP("v_min_u32_dpp", v, "row_mirror");
  unsigned r0, r1;
  swap_rows<false>(v, r0, r1);
  v = r0 < r1 ? r0 : r1;
  swap_rows<true>(v, r0, r1);
  return r0 < r1 ? r0 : r1;
}

// Among the lanes of `tie` (all holding the maximum), the one the reference's reduction tree
// keeps: smallest key.  Rare path (exact ties of fp32 distances).
__device__ __forceinline__ int wave_tie_break(unsigned long long tie, int idx, int log2bs) {
  const bool in = (tie >> lane_id()) & 1ull;
  const unsigned key = in ? fps_key(idx, log2bs) : 0xFFFFFFFFu;
  const unsigned mk = wave_min_u32(key);
  return __builtin_ctzll(__ballot(key == mk));
}

// Lane (wave-uniform) holding the best candidate under the reference's order: largest value,
// ties broken by the smallest key.  The common case (a unique maximum) needs one float
// reduction and one ballot; only real ties pay for the key reduction.  LANES < 64: only the
// first LANES lanes hold candidates.
template <int LANES = 64>
__device__ __forceinline__ int wave_argmax_lane(float v, int idx, int log2bs) {
  const float m = wave_max_f32<LANES>(v);
  unsigned long long tie = __ballot(v == m);
  if (LANES < 64) tie &= (1ull << LANES) - 1ull;
  if (__popcll(tie) <= 1) return tie ? __builtin_ctzll(tie) : 0;
  return wave_tie_break(tie, idx, log2bs);
}

struct FpsPick { int idx; float x, y, z; };

// After the barrier: wave w's candidate sits in slot[w*8 .. w*8+4] = (value, idx, x, y, z); every
// lane returns the workgroup's pick (largest value, ties by the reference's key).
template <int NW>
__device__ __forceinline__ FpsPick fps_pick_collect(const float *slot, int log2bs) {
  const int lane = lane_id();
  float sv = -2.0f, sx = 0.f, sy = 0.f, sz = 0.f;  // -2 < every real candidate (>= -1)
  int si = 0;
  if (lane < NW) {
    const float4 a = *reinterpret_cast<const float4 *>(slot + lane * 8);
    sv = a.x; si = __builtin_bit_cast(int, a.y); sx = a.z; sy = a.w;
    sz = slot[lane * 8 + 4];
  }
  const int best = wave_argmax_lane<(NW <= 8 ? 8 : NW <= 16 ? 16 : 64)>(sv, si, log2bs);
  FpsPick p;
  p.idx = __builtin_amdgcn_readlane(si, best);
  p.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sx), best));
  p.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sy), best));
  p.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sz), best));
  return p;
}

// One candidate per lane (value v, point index idx, its coordinates) -> the workgroup's pick,
// known to every lane together with its coordinates (so the next round needs no dependent
// global load).  slot: NW x 8 floats of LDS for this round's parity; ONE barrier per round:
// a wave can only overwrite a parity buffer two rounds later, i.e. after every wave has
// passed the barrier that follows its reads of that buffer.
template <int NW>
__device__ __forceinline__ FpsPick fps_block_pick(float v, int idx, float x, float y, float z,
                                                  float *slot, int log2bs) {
  const int lane = lane_id();
  const int w = threadIdx.x / kWave;
  const int win = wave_argmax_lane(v, idx, log2bs);
  if (lane == win) {
    float4 a = make_float4(v, __builtin_bit_cast(float, idx), x, y);
    *reinterpret_cast<float4 *>(slot + w * 8) = a;
    slot[w * 8 + 4] = z;
  }
  __syncthreads();
  return fps_pick_collect<NW>(slot, log2bs);
}

__device__ __forceinline__ bool fps_skipped(float x, float y, float z) {
  const float mag = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
  return (double)mag <= 1e-3;  // double compare: the reference's literal is a double
}


}  // namespace fps
