// 3dioumatch_amd/csrc/mlp_chain.hip -- the shared MLP of a set-abstraction module as a REGISTER
// CHAIN: conv(1x1) -> BatchNorm -> ReLU -> conv(1x1) (-> statistics / max over nsample) for one
// 32-column tile per wave without the intermediate activation ever leaving the registers
// (pointnet2/pytorch_utils.py:14-39,70-124 + the max-pool of pointnet2_modules.py:256-262).
//
// Why it chains without data movement (v_mfma_f32_32x32x16_bf16, wave64):
//   A operand: lane l holds A[m = l & 31][k = 8 (l >> 5) + e], e = 0..7
//   B operand: lane l holds B[k = 8 (l >> 5) + e][n = l & 31]
//   D        : lane l, register v holds D[(v & 3) + 8 (v >> 2) + 4 (l >> 5)][l & 31]
// * "N form"  D[channel][sample] = W (A) x act (B): a lane owns ONE sample column and 16 channels
//   per 32-channel block.  Those registers ARE an operand whose reduction axis is the channel
//   (k slot (l >> 5, e) <-> channel 16 t + 8 (e >> 2) + 4 (l >> 5) + (e & 3) for step t): the
//   next layer consumes them directly, the weight image being stored in the same k order.
// * "T form"  D[sample][channel] = act (A: m = sample) x W^T (B: n = channel): a lane owns ONE
//   channel and 16 samples per tile -- BatchNorm statistics and the max over nsample are
//   IN-LANE reductions (the forward GEMM's epilogue transposed its accumulators through LDS for
//   them: +45 % on the layer, profiles/r4_fwd_with_statistics.txt).
//   The operand registers are the same in both forms; only the MFMA's argument order differs.
// So: layer 2 in N form -> BN + ReLU in registers -> layer 3 in T form -> statistics, extrema
// and the y3 store from the accumulators.  Layer 2's own statistics need a pass of their own
// (they must exist before its ReLU): the same kernel without layer 3, in T form, nothing stored.
//
// fp32 products as six bf16 MFMAs on the exact three-term split (mlp_operand.h).  The weights are
// split ONCE per workgroup into fragment-ordered bf16 images in LDS (one ds_read_b128 per term
// and fragment, conflict-free; 72 KB for 64x64 + 128x64), two workgroups per CU.
//
// Input modes of the first operand (the activation that enters layer 2):
//   LIN4: relu(bn(W1 . x4)) recomputed from the 4-channel network input (SA1: the 4 -> 64 layer
//         is virtual, mlp_first4.hip); per element four FMAs + the BatchNorm FMA, same bits as
//         every other kernel that recomputes it (lin4).
#include "common.h"
#include "mlp_operand.h"
#include <stdlib.h>
#include <mutex>
#include <set>
#include <type_traits>

namespace {

struct ChainArgs {
  int r;                 // columns per cloud (multiple of 64)
  int tiles_per_cloud;   // r / 32
  const float *x4;       // (b, 4, r)
  const char *wimg;      // the weight images + layer-1 tables chain_prep_kernel left (kImgBytes)
  const float *sc2, *sh2;  // (64) BatchNorm of layer 2 (full pass)
  const float *gamma3;   // (M3): sign decides which extreme wins the pool
  float *y2;             // (b, 64, r) or null
  float *y3;             // (b, M3, r) or null
  float *pairs;          // (workgroups, C, 2): (mean, M2) of the workgroup's 4 * kTilesPerWave * 32
                         // columns, C = 64 (stats pass) / M3
  float *ext;            // 2 planes of (b, M3, r / NS): winning raw value, first index
  size_t ext_plane;
  int store_mode;        // bit 0: y2 with streaming stores, bit 1: y3 with streaming stores
};

constexpr int kChainK = 64;  // channels entering layer 2 / layer 3 (SA1: 64 -> 64 -> 128)

// byte offset of fragment (term, step t, half h, row) in a weight image of R rows
template <int R>
__device__ __forceinline__ int img_off(int term, int t, int h, int row) {
  return (((term * (kChainK / 16) + t) * 2 + h) * R + row) * 16;
}

__device__ __forceinline__ Split3 lds_frag(const char *img, int term_stride) {
  Split3 s;
  s.hi = *reinterpret_cast<const bf16x8 *>(img);
  s.mid = *reinterpret_cast<const bf16x8 *>(img + term_stride);
  s.lo = *reinterpret_cast<const bf16x8 *>(img + 2 * term_stride);
  return s;
}

// acc[i] += w[i] x s (N form: the weight is the A operand) or s x w[i] (T form), i < NB: the six
// significant partial products, small ones first, the NB independent accumulators interleaved
template <int NB, bool TFORM, bool FIRST = false>
__device__ __forceinline__ void mfma6(f32x16 (&acc)[NB], const Split3 &s, const Split3 (&w)[NB]) {
  // FIRST: the accumulators are not read -- the first product starts from a literal zero (an inline
  // constant of the MFMA: no register is cleared; 96 v_mov per tile otherwise)
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define CH_STEP(WT, ST, C0)                                                                        \
  _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                  \
    acc[i] = TFORM ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.ST, w[i].WT, (C0) ? zero : acc[i], 0, 0, 0) \
                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i].WT, s.ST, (C0) ? zero : acc[i], 0, 0, 0)
  CH_STEP(lo, hi, FIRST);
  CH_STEP(hi, lo, false);
  CH_STEP(mid, mid, false);
  CH_STEP(mid, hi, false);
  CH_STEP(hi, mid, false);
  CH_STEP(hi, hi, false);
#undef CH_STEP
}

// running shifted sums of one channel (a lane's 16 samples per tile)
struct RunStat { float shift, s1, s2; };

// FULL = false: statistics of layer 2's output only (T form, nothing stored)
// FULL = true : layer 2 (N form, y2 stored) -> BN + ReLU -> layer 3 (T form): y3 stored, its
//               statistics and, per group of NS columns, the extremum that wins the max-pool
// A workgroup = 4 waves x kTilesPerWave tiles of 32 columns (one pooling group of 64 per wave).
// MANY small workgroups, handed out by the hardware dispatcher: a persistent launch with one fixed
// share per resident workgroup ends when its unluckiest workgroup does -- inside the train step a
// few CUs are busy with the side stream's sampling kernels, the workgroups that could not become
// resident ran after all others and the launch took twice as long (measured); a single-word work
// queue costs ~10 k serialized atomics per launch (90 per microsecond: slower than no chaining).
constexpr int kTilesPerWave = 2;
constexpr int kW2Bytes = 3 * (kChainK / 16) * 2 * 64 * 16;    // 24 576
constexpr int kW3Bytes = 3 * (kChainK / 16) * 2 * 128 * 16;   // 49 152
constexpr int kTabBytes = 64 * 16 + 64 * 8;                   // w1 rows (float4) + (sc1, sh1)
constexpr int kImgBytes = kW2Bytes + kW3Bytes + kTabBytes;    // global image: [W2][W3T][tables]

// One launch per forward: the weights as fragment-ordered bf16 images (exact three-term split)
// and the first layer's tables, in the order the passes hold them in LDS.
__global__ void __launch_bounds__(256)
chain_prep_kernel(const float *__restrict__ w1, const float *__restrict__ sc1, const float *__restrict__ sh1,
                  const float *__restrict__ w2, const float *__restrict__ w3, char *__restrict__ img) {
  constexpr int T = kChainK / 16;
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g < 64 * T * 2) {
    // W2 as the A operand of the N form / the B operand of the T form (the same registers): row =
    // output channel, k slot (t, h, e) <-> input channel 16 t + 8 h + e
    const int row = g / (T * 2), th = g % (T * 2), t = th >> 1, hh = th & 1;
    const float *src = w2 + (size_t)row * kChainK + 16 * t + 8 * hh;
    const Split3 s = split3(*reinterpret_cast<const float4 *>(src), *reinterpret_cast<const float4 *>(src + 4));
    char *dst = img + img_off<64>(0, t, hh, row);
    *reinterpret_cast<bf16x8 *>(dst) = s.hi;
    *reinterpret_cast<bf16x8 *>(dst + kW2Bytes / 3) = s.mid;
    *reinterpret_cast<bf16x8 *>(dst + 2 * (kW2Bytes / 3)) = s.lo;
  } else if (g < (64 + 128) * T * 2) {
    // W3 as the B operand of the T form: row = output channel, k slot (t, h, e) <-> input channel
    // 16 t + 8 (e >> 2) + 4 h + (e & 3) -- the order layer 2's accumulators hold them in
    const int q = g - 64 * T * 2;
    const int row = q / (T * 2), th = q % (T * 2), t = th >> 1, hh = th & 1;
    const float *src = w3 + (size_t)row * kChainK + 16 * t + 4 * hh;
    const Split3 s = split3(*reinterpret_cast<const float4 *>(src), *reinterpret_cast<const float4 *>(src + 8));
    char *dst = img + kW2Bytes + img_off<128>(0, t, hh, row);
    *reinterpret_cast<bf16x8 *>(dst) = s.hi;
    *reinterpret_cast<bf16x8 *>(dst + kW3Bytes / 3) = s.mid;
    *reinterpret_cast<bf16x8 *>(dst + 2 * (kW3Bytes / 3)) = s.lo;
  } else if (g < (64 + 128) * T * 2 + 64) {
    const int ch = g - (64 + 128) * T * 2;
    reinterpret_cast<float4 *>(img + kW2Bytes + kW3Bytes)[ch] = *reinterpret_cast<const float4 *>(w1 + (size_t)ch * 4);
    reinterpret_cast<float2 *>(img + kW2Bytes + kW3Bytes + 1024)[ch] = make_float2(sc1[ch], sh1[ch]);
  }
}

template <int M3B, int NS, bool FULL>
__global__ void __launch_bounds__(256, FULL ? 2 : 4) chain_lin4_kernel(const ChainArgs a) {
  constexpr int M2 = 64, M3 = 32 * M3B, T = kChainK / 16;
  static_assert(M3 == 128, "the image layout of chain_prep_kernel");
  constexpr int W2_BYTES = kW2Bytes, W3_BYTES = FULL ? kW3Bytes : 0;
  constexpr int W2_TERM = kW2Bytes / 3, W3_TERM = kW3Bytes / 3;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *w2img = lds, *w3img = lds + W2_BYTES;
  float4 *w1tab = reinterpret_cast<float4 *>(lds + W2_BYTES + W3_BYTES);          // [64]
  float2 *c1tab = reinterpret_cast<float2 *>(lds + W2_BYTES + W3_BYTES + 1024);   // [64] sc1, sh1
  float2 *c2tab = reinterpret_cast<float2 *>(lds + W2_BYTES + W3_BYTES + 1536);   // [64] sc2, sh2

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- the prepared images -> LDS (16-byte pieces; the statistics pass skips the W3 image)
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(a.wimg);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    constexpr int HEAD = (kW2Bytes + (FULL ? kW3Bytes : 0)) / 16, TAB = kTabBytes / 16;
    for (int i = tid; i < HEAD; i += 256) dst[i] = src[i];
    if (tid < TAB) dst[HEAD + tid] = src[(kW2Bytes + kW3Bytes) / 16 + tid];
    if constexpr (FULL) {
      if (tid >= 64 && tid < 128) c2tab[tid - 64] = make_float2(a.sc2[tid - 64], a.sh2[tid - 64]);
    }
  }
  __syncthreads();

  // lane-constant bases: every table / fragment access below is one base register + an immediate
  const float4 *w1h = w1tab + 8 * h;
  const float2 *c1h = c1tab + 8 * h;
  const float2 *c2h = c2tab + 4 * h;
  const char *w2lane = w2img + (h * M2 + l31) * 16;
  const char *w3lane = w3img + (h * M3 + l31) * 16;
  constexpr int CB = FULL ? M3B : 2;  // channel blocks whose statistics this pass leaves
  RunStat st[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) st[j] = {0.f, 0.f, 0.f};
  float sgn[FULL ? M3B : 1];
  if constexpr (FULL) {
#pragma unroll
    for (int j = 0; j < M3B; ++j) sgn[j] = a.gamma3[32 * j + l31] < 0.f ? -1.f : 1.f;
  }
  constexpr int TPG = NS > 32 ? NS / 32 : 1;  // tiles per pooling group
  float best[M3B][NS == 16 ? 2 : 1];
  int at[M3B][NS == 16 ? 2 : 1];

  // this lane's column of the 4-channel input, one tile ahead
  auto load_x = [&](int tile, float (&x)[4]) {
    const int b = tile / a.tiles_per_cloud, col = (tile - b * a.tiles_per_cloud) * 32 + l31;
    const float *p = a.x4 + (size_t)b * 4 * a.r + col;
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = p[(size_t)c * a.r];
  };
  const int tile0 = ((int)blockIdx.x * 4 + wave) * kTilesPerWave;
  constexpr int tiles_here = kTilesPerWave;
  int done_tiles = 0;
  {
  float xn[4];
  load_x(tile0, xn);

#pragma unroll 1
  for (int it = 0; it < tiles_here; ++it, ++done_tiles) {
    const int tile = tile0 + it;
    const int b = tile / a.tiles_per_cloud, col0 = (tile - b * a.tiles_per_cloud) * 32;
    float x[4] = {xn[0], xn[1], xn[2], xn[3]};
    if (it + 1 < tiles_here) load_x(tile + 1, xn);

    // ---- layer 2 over the 64 recomputed channels of layer 1.  Software pipeline inside the wave:
    // while the matrix pipe works on step t (12 MFMAs, 32 cycles each) the wave prepares the
    // operand of step t+1 (LDS reads first, then the vector work) -- two co-resident waves run
    // the same stream nearly in lock-step and do not hide it for each other (measured: without
    // the interleave a tile cost MFMA time + VALU time, 12 k cycles).
    f32x16 acc2[2];
    auto prep_a1 = [&](int t) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float4 w = w1h[16 * t + e];
        const float2 c = c1h[16 * t + e];
        v[e] = fmaxf(__fmaf_rn(lin4(w, x[0], x[1], x[2], x[3]), c.x, c.y), 0.f);
      }
      return split3(v);
    };
    auto load_w2 = [&](int t, Split3 (&w)[2]) {
#pragma unroll
      for (int i = 0; i < 2; ++i) w[i] = lds_frag(w2lane + img_off<M2>(0, t, 0, 32 * i), W2_TERM);
    };
    {
      Split3 sc = prep_a1(0), wc[2];
      load_w2(0, wc);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        Split3 sn = sc, wn[2] = {wc[0], wc[1]};
        if (t + 1 < T) {
          sn = prep_a1(t + 1);
          load_w2(t + 1, wn);
        }
        if (t == 0) mfma6<2, !FULL, true>(acc2, sc, wc);
        else mfma6<2, !FULL>(acc2, sc, wc);
        if (t + 1 < T) {
          // 12 MFMAs over 22 LDS reads + ~92 vector instructions: reads under the first MFMAs
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
          }
        }
        sc = sn; wc[0] = wn[0]; wc[1] = wn[1];
        __builtin_amdgcn_sched_barrier(0);  // (one scheduling region per step)
      }
    }

    // statistics of a T-form block set: lane <-> channel, 16 samples per lane and tile
    auto accumulate = [&](auto &blocks, int nb) {
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (j >= nb) break;
        if (done_tiles == 0) st[j].shift = blocks[j][0];
        float s1 = st[j].s1, s2 = st[j].s2;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float d = blocks[j][q] - st[j].shift;
          s1 += d;
          s2 = __fmaf_rn(d, d, s2);
        }
        st[j].s1 = s1; st[j].s2 = s2;
      }
    };

    if constexpr (!FULL) {
      accumulate(acc2, 2);
    } else {
      // y2 (N form): register q of block i = channel 32 i + (q & 3) + 8 (q >> 2) + 4 h, this
      // lane's column -- 128-byte row segments per half-wave
      if (a.y2 != nullptr) {
        float *dst = a.y2 + ((size_t)b * M2 + 4 * h) * a.r + col0 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            float *d1 = &dst[(size_t)(32 * i + (q & 3) + 8 * (q >> 2)) * a.r];
            if (a.store_mode & 1) __builtin_nontemporal_store(acc2[i][q], d1);
            else *d1 = acc2[i][q];
          }
      }
      // ---- layer 3 (T form) on relu(bn(y2)) taken straight from layer 2's accumulators; the four
      // channel blocks as two halves of 12 MFMAs, the other half's weight fragments and the next
      // step's operand prepared underneath
      f32x16 acc3[M3B];
      static_assert(M3B == 4, "two halves of two blocks");
      auto prep_a2 = [&](int t) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float2 c = c2h[16 * t + 8 * (e >> 2) + (e & 3)];
          v[e] = fmaxf(__fmaf_rn(acc2[t >> 1][8 * (t & 1) + e], c.x, c.y), 0.f);
        }
        return split3(v);
      };
      auto load_w3 = [&](int t, int j0, Split3 (&w)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) w[i] = lds_frag(w3lane + img_off<M3>(0, t, 0, 32 * (j0 + i)), W3_TERM);
      };
      {
        f32x16 lo2[2], hi2[2];
        Split3 sc = prep_a2(0), wa[2], wb[2];
        load_w3(0, 0, wa);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          load_w3(t, 2, wb);
          Split3 sn = sc;
          if (t + 1 < T) sn = prep_a2(t + 1);
          if (t == 0) mfma6<2, true, true>(lo2, sc, wa);
          else mfma6<2, true>(lo2, sc, wa);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < T) load_w3(t + 1, 0, wa);
          if (t == 0) mfma6<2, true, true>(hi2, sc, wb);
          else mfma6<2, true>(hi2, sc, wb);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
          sc = sn;
          __builtin_amdgcn_sched_barrier(0);
        }
        acc3[0] = lo2[0]; acc3[1] = lo2[1]; acc3[2] = hi2[0]; acc3[3] = hi2[1];
      }
      // y3: lane = channel 32 j + l31, registers 4 g .. 4 g + 3 = columns 8 g + 4 h + (0..3)
      if (a.y3 != nullptr) {
#pragma unroll
        for (int j = 0; j < M3B; ++j) {
          float *dst = a.y3 + ((size_t)b * M3 + 32 * j + l31) * a.r + col0 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 o = {acc3[j][4 * g], acc3[j][4 * g + 1], acc3[j][4 * g + 2], acc3[j][4 * g + 3]};
            if (a.store_mode & 2) __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(dst + 8 * g));
            else *reinterpret_cast<f32x4 *>(dst + 8 * g) = o;
          }
        }
      }
      accumulate(acc3, M3B);
      // ---- the extremum that wins the max-pool (relu(y*sc + sh) is monotone in y; the sign of
      // gamma says which way), first occurrence: the lane scans its samples in increasing order
      const int p = it % TPG;  // tile within its group (NS == 64: two tiles per group)
#pragma unroll
      for (int j = 0; j < M3B; ++j) {
        if (p == 0) {
#pragma unroll
          for (int gq = 0; gq < (NS == 16 ? 2 : 1); ++gq) { best[j][gq] = -__builtin_inff(); at[j][gq] = 0; }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int gq = NS == 16 ? (q >> 3) : 0;
          const int smp = (q & 3) + 8 * (q >> 2) + 4 * h;           // sample within the tile
          const int idx = NS == 16 ? smp - 16 * gq : smp + 32 * p;  // ... within its group
          const float tv = acc3[j][q] * sgn[j];
          if (tv > best[j][gq]) { best[j][gq] = tv; at[j][gq] = idx; }
        }
      }
      if (p == TPG - 1) {
        const int groups = a.r / NS;
        const int g0 = NS == 16 ? col0 / 16 : (col0 - 32 * p) / NS;
        int *ei = reinterpret_cast<int *>(a.ext);
#pragma unroll
        for (int j = 0; j < M3B; ++j)
#pragma unroll
          for (int gq = 0; gq < (NS == 16 ? 2 : 1); ++gq) {
            float bv = best[j][gq];
            int ba = at[j][gq];
            const float ob = __shfl_xor(bv, 32, kWave);
            const int oa = __shfl_xor(ba, 32, kWave);
            if (ob > bv || (ob == bv && oa < ba)) { bv = ob; ba = oa; }
            if (h == 0) {
              const size_t o = ((size_t)b * M3 + 32 * j + l31) * groups + g0 + gq;
              a.ext[o] = bv * sgn[j];
              ei[a.ext_plane + o] = ba;
            }
          }
      }
    }
  }

  }

  // ---- (mean, M2) per channel: the two half-waves hold different samples of the same channel, the
  // four waves different columns (equal counts everywhere); one pair per workgroup and channel
  __syncthreads();  // every wave is done with the images: their space takes the waves' pairs
  float2 *wstat = reinterpret_cast<float2 *>(lds);  // [4 waves][32 * CB channels]
  const float n_half = 16.f * (float)kTilesPerWave;
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const float mh = st[j].shift + st[j].s1 / n_half;
    const float qh = fmaxf(st[j].s2 - st[j].s1 * st[j].s1 / n_half, 0.f);
    const float mo = __shfl_xor(mh, 32, kWave), qo = __shfl_xor(qh, 32, kWave);
    if (h == 0) {
      const float dlt = mo - mh;
      // n_a n_b / (n_a + n_b) = n_half / 2
      wstat[wave * (32 * CB) + 32 * j + l31] = make_float2(0.5f * (mh + mo), (qh + qo) + 0.5f * n_half * dlt * dlt);
    }
  }
  __syncthreads();
  if (tid < 32 * CB) {
    float mean = 0.f, m2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mean += wstat[w * (32 * CB) + tid].x; m2 += wstat[w * (32 * CB) + tid].y; }
    mean *= 0.25f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float d = wstat[w * (32 * CB) + tid].x - mean;
      m2 = __fmaf_rn(2.f * n_half * d, d, m2);
    }
    float *dst = a.pairs + ((size_t)blockIdx.x * (32 * CB) + tid) * 2;
    dst[0] = mean;
    dst[1] = m2;
  }
}

int chain_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}

bool chain_shape_ok(int b, int r, int ns) {
  return b > 0 && r > 0 && r % (32 * 4 * kTilesPerWave) == 0 && (ns == 16 || ns == 32 || ns == 64) &&
         r % ns == 0 && (long long)b * (r / 32) < (1LL << 30);
}

// workgroups of a pass: 4 waves x kTilesPerWave tiles each
int chain_workgroups(int b, int r) { return (int)((long long)b * (r / 32) / (4 * kTilesPerWave)); }

bool chain_enabled() {
  static const bool off = getenv("MLP_CHAIN_FWD") && atoi(getenv("MLP_CHAIN_FWD")) == 0;
  return !off;
}

template <typename Kern>
void chain_launch(Kern kern, int wgs, size_t lds_bytes, hipStream_t stream, const ChainArgs &args) {
  // (every instantiation has the same pointer TYPE: remember the kernels by address)
  static std::mutex mu;
  static std::set<const void *> done;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (done.insert(reinterpret_cast<const void *>(kern)).second)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  }
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds_bytes, stream, args);
}

// ---- BatchNorm coefficients from the workgroups' equal-count (mean, M2) pairs (part, channel, 2):
// one wave per channel, sums in double around the first part's mean (no cancellation):
// mean = ref + s1 / P, M2 = sum M2_i + n_part * (s2 - s1^2 / P)
__device__ __forceinline__ double chain_wave_sum(double v) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, off, kWave);
    hi = __shfl_xor(hi, off, kWave);
    v += __hiloint2double(hi, lo);
  }
  return v;
}

__global__ void __launch_bounds__(256)
chain_finalize_kernel(int c, int parts, int n_part, const float *__restrict__ pairs,
                      const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                      float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
                      float *__restrict__ mean_out, float *__restrict__ invstd_out,
                      float *__restrict__ scale_out, float *__restrict__ shift_out) {
  __shared__ double red[4][3];
  const int ch = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
  const double ref = (double)pairs[(size_t)ch * 2];
  double s1 = 0.0, s2 = 0.0, sm2 = 0.0;
  const float *base = pairs + (size_t)ch * 2;
  int p = tid;
  for (; p + 3 * 256 < parts; p += 4 * 256) {  // four independent loads in flight per lane
    float2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float2 *>(base + (size_t)(p + u * 256) * c * 2);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double d = (double)v[u].x - ref;
      s1 += d; s2 += d * d; sm2 += (double)v[u].y;
    }
  }
  for (; p < parts; p += 256) {
    const float2 v = *reinterpret_cast<const float2 *>(base + (size_t)p * c * 2);
    const double d = (double)v.x - ref;
    s1 += d; s2 += d * d; sm2 += (double)v.y;
  }
  s1 = chain_wave_sum(s1); s2 = chain_wave_sum(s2); sm2 = chain_wave_sum(sm2);
  if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; red[wave][2] = sm2; }
  __syncthreads();
  if (tid != 0) return;
  s1 = red[0][0] + red[1][0] + red[2][0] + red[3][0];
  s2 = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  sm2 = red[0][2] + red[1][2] + red[2][2] + red[3][2];
  const double P = (double)parts, n = P * (double)n_part;
  const double mean = ref + s1 / P;
  double m2 = sm2 + (double)n_part * (s2 - s1 * s1 / P);
  if (m2 < 0.0) m2 = 0.0;
  const double var = m2 / n;  // biased, used for normalisation
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float fmean = (float)mean;
  mean_out[ch] = fmean;
  invstd_out[ch] = invstd;
  const float sc = gamma[ch] * invstd;
  scale_out[ch] = sc;
  shift_out[ch] = beta[ch] - fmean * sc;
  if (running_mean != nullptr) {  // nn.BatchNorm: unbiased variance in the running estimate
    const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * fmean;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
  }
}

}  // namespace

#define MLP_API extern "C" __attribute__((visibility("default")))

// Number of (mean, M2) pairs per channel a pass of the chained forward of a 4 -> 64 -> 64 -> m3
// module leaves (one per workgroup) and the columns each covers; 0 when the shape is outside the
// kernel (m3 = 128, nsample 16 / 32 / 64, r a multiple of 256).
MLP_API int mlp_chain_lin4_parts(int b, int r, int m3, int ns, int *cols_per_part) {
  if (!chain_enabled() || m3 != 128 || !chain_shape_ok(b, r, ns)) return 0;
  if (cols_per_part) *cols_per_part = 32 * 4 * kTilesPerWave;
  return chain_workgroups(b, r);
}

// bytes of the weight-image scratch mlp_chain_lin4_prepare fills
MLP_API size_t mlp_chain_lin4_image_bytes(void) { return (size_t)kImgBytes; }

// The weights of the module as fragment-ordered bf16 images (exact three-term split) + the first
// layer's tables -> img (mlp_chain_lin4_image_bytes(), 16-byte aligned); once per forward, before
// the two passes.  w1 (64,4), sc1 / sh1 (64), w2 (64,64), w3 (128,64).
MLP_API int mlp_chain_lin4_prepare(const float *w1, const float *sc1, const float *sh1, const float *w2,
                                   const float *w3, void *img, void *stream_) {
  if (!w1 || !sc1 || !sh1 || !w2 || !w3 || !img || (reinterpret_cast<size_t>(w1) & 15) ||
      (reinterpret_cast<size_t>(w2) & 15) || (reinterpret_cast<size_t>(w3) & 15) ||
      (reinterpret_cast<size_t>(img) & 15))
    return (int)hipErrorInvalidValue;
  constexpr int items = (64 + 128) * (kChainK / 16) * 2 + 64;
  hipLaunchKernelGGL(chain_prep_kernel, dim3((items + 255) / 256), dim3(256), 0, (hipStream_t)stream_, w1,
                     sc1, sh1, w2, w3, (char *)img);
  return pn2_launch_status();
}

// Statistics pass: (mean, M2) pairs2 (parts, 64, 2) of y2 = w2 . relu(bn1(w1 . x4)) -- nothing else
// is written.  img: what mlp_chain_lin4_prepare left.
MLP_API int mlp_chain_lin4_stats(int b, int r, int ns, const float *x4, const void *img, float *pairs2,
                                 void *stream_) {
  if (!chain_shape_ok(b, r, ns) || !x4 || !img || !pairs2) return (int)hipErrorInvalidValue;
  ChainArgs a = {};
  a.r = r; a.tiles_per_cloud = r / 32;
  a.x4 = x4; a.wimg = (const char *)img; a.pairs = pairs2;
  const size_t lds_bytes = kW2Bytes + kTabBytes + 512;
  chain_launch(chain_lin4_kernel<4, 32, false>, chain_workgroups(b, r), lds_bytes, (hipStream_t)stream_, a);
  return pn2_launch_status();
}

// Full pass: y2 (b,64,r) and y3 (b,128,r) stored (either may be null), pairs3 (parts,128,2), ext =
// 2 planes of (b,128,r/ns) as mlp_gemm_forward_stats_pool leaves them.  sc2 / sh2: layer 2's
// BatchNorm (from the statistics pass), gamma3: layer 3's BatchNorm weight.
MLP_API int mlp_chain_lin4_forward(int b, int r, int ns, const float *x4, const void *img, const float *sc2,
                                   const float *sh2, const float *gamma3, float *y2, float *y3,
                                   float *pairs3, float *ext, void *stream_) {
  if (!chain_shape_ok(b, r, ns) || !x4 || !img || !sc2 || !sh2 || !gamma3 || !pairs3 || !ext ||
      (y3 && (reinterpret_cast<size_t>(y3) & 15)))
    return (int)hipErrorInvalidValue;
  ChainArgs a = {};
  a.r = r; a.tiles_per_cloud = r / 32;
  a.x4 = x4; a.wimg = (const char *)img; a.sc2 = sc2; a.sh2 = sh2;
  a.gamma3 = gamma3; a.y2 = y2; a.y3 = y3; a.pairs = pairs3; a.ext = ext;
  a.ext_plane = (size_t)b * 128 * (r / ns);
  a.store_mode = getenv("MLP_CHAIN_STORE") ? atoi(getenv("MLP_CHAIN_STORE")) : 1;
  const size_t lds_bytes = kImgBytes + 512;
  hipStream_t stream = (hipStream_t)stream_;
  const int wgs = chain_workgroups(b, r);
  if (ns == 16) chain_launch(chain_lin4_kernel<4, 16, true>, wgs, lds_bytes, stream, a);
  else if (ns == 32) chain_launch(chain_lin4_kernel<4, 32, true>, wgs, lds_bytes, stream, a);
  else chain_launch(chain_lin4_kernel<4, 64, true>, wgs, lds_bytes, stream, a);
  return pn2_launch_status();
}

// Training-mode BatchNorm coefficients (and running statistics) of c channels from a pass's
// equal-count pairs (parts, c, 2), n_part columns each.
MLP_API int mlp_chain_finalize(int c, int parts, int n_part, const float *pairs, const float *gamma,
                               const float *beta, float eps, float momentum, float *running_mean,
                               float *running_var, float *mean, float *invstd, float *scale, float *shift,
                               void *stream_) {
  if (c <= 0 || parts <= 0 || n_part <= 0 || !pairs || !gamma || !beta || !mean || !invstd || !scale || !shift)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(chain_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream_, c, parts, n_part,
                     pairs, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  return pn2_launch_status();
}
