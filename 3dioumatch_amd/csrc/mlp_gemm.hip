// 3dioumatch_amd/csrc/mlp_gemm.hip -- the grouped shared-MLP contraction on the matrix cores
// (gfx950, v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation), with the BatchNorm /
// ReLU algebra folded into the operand loads.
//
// What it replaces: nn.Conv2d(1x1) of a shared-MLP layer (pointnet2/pytorch_utils.py:70-124)
// and its autograd backward, i.e. for activations X (B, K, R) and a weight W (M, K):
//     forward : Y[b]  = W * X[b]                                   (M x R)
//     dgrad   : dX[b] = W^T * dY[b]                                (K x R)
//     wgrad   : dW    = sum_b dY[b] * X[b]^T                       (M x K)
// R = npoint*nsample is the contiguous axis of every tensor, so all three are "R-contiguous"
// GEMMs.  The twist is in what the operands ARE.  A layer's input is relu(bn(Y_prev)) and the
// gradient that enters its convolution is the BatchNorm/ReLU backward of dZ; instead of
// materialising those tensors (one write + one read of a GB-scale activation each), the
// operand tiles are transformed on their way from HBM into LDS:
//     OP_DIRECT  : x
//     OP_BNRELU  : max(x*scale[k] + shift[k], 0)                    (input of the next layer)
//     OP_DY      : a[k]*(([y*sc+sh > 0] ? dz : 0) - c1[k] - ((y-mu[k])*is[k])*c2[k])
//                  from the pair (y, dz)                            (BN+ReLU backward)
//     OP_POOLDY  : the same with dz[k][g*ns+s] = [s == argmax[k][g]] * dpooled[k][g]: the
//                  backward of max-over-nsample + ReLU + BN of an SA module's last layer,
//                  from y and the two small (B,C,m) tensors -- dz/dy are never materialised
//
// Kernel shape: 256 lanes = 4 waves; a workgroup owns TM x TN of the output for one cloud;
// K is walked in chunks of 16 through LDS ([k][m] and [k][n], both unit-stride for the MFMA
// fragment reads); each wave accumulates (TM/WM) x (TN/WN) in 32x32 MFMA blocks.
#include "common.h"
#include "mlp_operand.h"
#include <mutex>
#include <stdlib.h>

namespace {

constexpr int KC = 16;  // K chunk staged in LDS

// C[b] (M x R, ldc = R) = A (M x K, row-major, lda) * op(B[b]) (K x R)
// A_TRANS: the A operand is given transposed in memory (A[gm][gk] = a[gk*lda + gm]) -- dgrad reads
// the weight as stored instead of a transposed copy; the lane<->element mapping of the A loads is
// swapped so that they stay coalesced.
template <int TM, int TN, int WM, int WN, int MODE, bool A_TRANS>
__global__ void __launch_bounds__(256)
gemm_nn_kernel(int m_total, int k_total, int r, const float *__restrict__ a, int lda, OperandB opb,
               float *__restrict__ c, size_t b_stride_in, size_t b_stride_out) {
  constexpr int MB = TM / WM / 32, NB = TN / WN / 32;
  // [k][m]; the transposing stores put lane (kk, mm) at bank (kk*LDA + mm) % 64: with LDA = TM + 4
  // the 16 x 4 lanes of a wave hit 64 distinct banks (TM + 1 folded them onto 19)
  constexpr int LDA = TM + 4;
  __shared__ float As[KC * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[KC * TN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r0 = blockIdx.x * TN, m0 = blockIdx.y * TM, b = blockIdx.z;
  OperandB op = opb;
  const size_t in_off = (size_t)b * b_stride_in;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // Software pipeline: the global loads of chunk i+1 are issued before the MFMAs of chunk i and
  // only waited for (and transformed) when they are stored to LDS, so HBM/L2 latency hides
  // under the matrix pipe.
  constexpr int AE = TM * KC / 256;  // A elements per lane
  constexpr int SEG = TN / 16;       // B elements per lane: 16 lanes share one row
  const int bkk = tid >> 4, bnn = (tid & 15) * SEG;
  const bool vec_ok = (r & 3) == 0;
  float areg[AE], bx[SEG], bdz[SEG];
  RowCoef rc;
  bool brow_ok = false;
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      const int t = tid + e * 256;
      const int kk = A_TRANS ? t / TM : t % KC, mm = A_TRANS ? t % TM : t / KC;
      const int gm = m0 + mm, gk = k0 + kk;
      const size_t at = A_TRANS ? (size_t)gk * lda + gm : (size_t)gm * lda + gk;
      areg[e] = (gm < m_total && gk < k_total) ? a[at] : 0.f;
    }
    const int gk = k0 + bkk;
    brow_ok = gk < k_total;
    rc = load_row_coef<MODE>(op, gk, brow_ok);
    load_raw_segment<MODE, SEG>(op, in_off + (size_t)gk * r + r0 + bnn, r0 + bnn, r, vec_ok,
                                brow_ok, bx, bdz, b * k_total + gk);
  };
  auto stash = [&]() {
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      const int t = tid + e * 256;
      As[(A_TRANS ? t / TM : t % KC) * LDA + (A_TRANS ? t % TM : t / KC)] = areg[e];
    }
#pragma unroll
    for (int i = 0; i < SEG; i += 4) {
      float4 v;
      v.x = (brow_ok && r0 + bnn + i + 0 < r) ? transform<MODE>(bx[i + 0], bdz[i + 0], rc) : 0.f;
      v.y = (brow_ok && r0 + bnn + i + 1 < r) ? transform<MODE>(bx[i + 1], bdz[i + 1], rc) : 0.f;
      v.z = (brow_ok && r0 + bnn + i + 2 < r) ? transform<MODE>(bx[i + 2], bdz[i + 2], rc) : 0.f;
      v.w = (brow_ok && r0 + bnn + i + 3 < r) ? transform<MODE>(bx[i + 3], bdz[i + 3], rc) : 0.f;
      *reinterpret_cast<float4 *>(&Bs[bkk * TN + bnn + i]) = v;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < k_total; k0 += KC) {
    __syncthreads();  // the MFMAs of the previous chunk are done with LDS
    stash();
    __syncthreads();
    if (k0 + KC < k_total) fetch(k0 + KC);
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2) {
      const int krow = kk + (lane >> 5);
      float af[MB], bf[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) af[i] = As[krow * LDA + (wm * MB + i) * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < NB; ++j) bf[j] = Bs[krow * TN + (wn * NB + j) * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  // C/D layout of the 32x32 block: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
  float *cb = c + (size_t)b * b_stride_out;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = r0 + (wn * NB + j) * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + (wm * MB + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < m_total && col < r) __builtin_nontemporal_store(acc[i][j][q], &cb[(size_t)row * r + col]);
      }
    }
}

// ---- pipelined variant -------------------------------------------------------------------------
// Same tiles and operand transforms as gemm_nn_kernel, different inner machinery (the first
// kernel kept its matrix pipes 50-63 % busy: two barriers per 16-deep K chunk with the LDS stores
// between them, and an A tile fetched as eight bounds-checked single dwords per lane):
//   * LDS is double buffered: chunk i+1 is stored while chunk i is being multiplied -- ONE
//     barrier per chunk, and the stores sit in the middle of the MFMA stream;
//   * the A tile (weights) is fetched as 16-byte pieces with one row predicate each (two per lane
//     and chunk), or -- rows that are not 16-byte aligned, K = 131 / 259 -- as range-checked
//     BUFFER dwords (hardware bounds check instead of per-element exec masks); only the last,
//     partial K chunk takes the masked dword path;
//   * the loads of chunk i+2 are issued right after chunk i+1 left the registers.
// Requires r % TN == 0 and r % 4 == 0 (every shape of the network); other shapes use the kernel
// above.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// sum over the 32 lanes of each half-wave; the total lands in lanes 16..31 / 48..63
__device__ __forceinline__ float half_wave_sum(float v) {
#define HW_STEP(CTRL, RM) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xf, false))
  HW_STEP(0xB1, 0xf);   // quad_perm [1,0,3,2]
  HW_STEP(0x4E, 0xf);   // quad_perm [2,3,0,1]
  HW_STEP(0x141, 0xf);  // row_half_mirror
  HW_STEP(0x140, 0xf);  // row_mirror        -> every lane: the total of its row of 16
  HW_STEP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3 -> the total of the half-wave
#undef HW_STEP
  return v;
}

// the same butterfly with max / min: lanes 16..31 (48..63) end up with the extreme of lanes
// 0..31 (32..63); masked-out rows keep their own value (old = v)
template <bool MAXOP>
__device__ __forceinline__ float half_wave_extreme(float v) {
#define HX_STEP(CTRL, RM)                                                                        \
  {                                                                                              \
    const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(                      \
        __builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, RM, 0xf, false));         \
    v = MAXOP ? fmaxf(v, o) : fminf(v, o);                                                       \
  }
  HX_STEP(0xB1, 0xf);
  HX_STEP(0x4E, 0xf);
  HX_STEP(0x141, 0xf);
  HX_STEP(0x140, 0xf);
  HX_STEP(0x142, 0xa);
#undef HX_STEP
  return v;
}

// A_VEC: the rows of A are 16-byte aligned (lda % 4 == 0 and an aligned base)
// STATS: the epilogue also reduces every output row over the tile's columns to a
//        (mean, M2) pair for the BatchNorm that follows (one per row, cloud and column tile, all
//        tiles hold TN columns): the statistics pass no longer re-reads y from HBM
// POOL:  (16, 32 or 64 = nsample) the epilogue also leaves, per output row and group of POOL
//        consecutive columns, the raw value that will win the max-pool over nsample and where
//        it first occurs: relu(y*sc + sh) is monotone in y for a given channel -- increasing
//        when gamma >= 0, decreasing otherwise (sc = gamma * invstd) -- so the winner is the
//        group's largest or smallest raw value, known before the statistics are.  The pooling
//        pass no longer re-reads y either.  ext: 2 planes (value, first index) of
//        (b, m_total, r / POOL).
template <int TM, int TN, int WM, int WN, int MODE, bool A_TRANS, bool A_VEC, bool STATS = false,
          int POOL = 0, bool X6 = false>
#ifndef MLP_X6_OCC
#define MLP_X6_OCC 3
#endif
#ifndef MLP_X6_OCC_256
#define MLP_X6_OCC_256 3
#endif
__global__ void __launch_bounds__(256, X6 ? (((MODE <= OP_BNRELU || MODE == OP_LIN4) && !A_TRANS) ? (TM <= 128 ? MLP_X6_OCC : MLP_X6_OCC_256) : 2) : (((MODE <= OP_BNRELU || MODE == OP_LIN4) && TM <= 128) ? 4 : 2))
gemm_nn2_kernel(int m_total, int k_total, int r, const float *__restrict__ a, int lda,
                unsigned a_bytes, OperandB opb, float *__restrict__ c, size_t b_stride_in,
                size_t b_stride_out, float *__restrict__ stats = nullptr, int stat_channels = 0,
                float *__restrict__ ext = nullptr, size_t ext_plane = 0,
                const float *__restrict__ pool_gamma = nullptr) {
  constexpr int MB = TM / WM / 32, NB = TN / WN / 32;
  // LDS layout "k-quads": the chunk's 16 k are four planes of [row][4 consecutive k], so that a
  // lane fetches the fragments of FOUR MFMA steps with one 16-byte read (m or n = its row, the
  // plane = 2 * (lane >> 5) + half of the chunk): 2 * (MB + NB) reads feed the chunk's
  // 8 * MB * NB MFMAs and need no address arithmetic (one base register per operand, immediate
  // offsets).  The previous [k][row] layout cost one 4-byte read per MFMA operand and step, with
  // a VALU add per pair of reads, and kept the matrix pipes 50-70 % busy -- what a stream of
  // MFMAs that waits for one LDS read per MFMA reaches in isolation (tools/micro/mfma_peak.py).
  // The MFMA consumes k in the order (t, 8 + t), t = 0..7, instead of (2t, 2t + 1): a
  // permutation of the summation order inside the chunk.
  // B rows carry 16 bytes of padding after every 8 rows: the 16 lanes that stage one k row write
  // rows 8 apart (SEG = 8) and would otherwise meet in the same banks.
  // A_TRANS (dgrad: the weight as stored) stages 4-byte pieces of rows 4 apart, which would land
  // in 4 of the 16 bank groups: there the rows are rotated inside their group of 16 by the group's
  // number (a_slot).  Measured on one box, forward / dgrad shapes of tools/gemm_bench.py:
  // [k][row] layout 1454-1463 / 1911-1916 us, k-quads 1416-1418 / 1963-1994, k-quads with the
  // rotation 1448-1452 / 1887-1907 -- hence the rotation for the transposed operand only.
  constexpr int PLA = TM * 4 + 16;              // floats per A plane (+16: planes start 16 banks apart)
  auto a_slot = [](int row) { return A_TRANS ? ((row & ~15) | ((row + (row >> 4)) & 15)) : row; };
  constexpr int PLB = TN * 4 + (TN / 8) * 4;    // floats per B plane
  constexpr int AV = TM * KC / 4 / 256;  // 16-byte A pieces per lane and chunk
  constexpr int SEG = TN / 16;      // B elements per lane: 16 lanes share one row
  static_assert(AV >= 1 && SEG % 4 == 0, "tile too small for the vector paths");
  static_assert(KC == 16, "four k-quads per chunk");
  __shared__ __attribute__((aligned(16))) float As[2][4 * PLA];
  __shared__ __attribute__((aligned(16))) float Bs[2][4 * PLB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r0 = blockIdx.x * TN, m0 = blockIdx.y * TM, b = blockIdx.z;
  OperandB op = opb;
  const size_t in_off = (size_t)b * b_stride_in;
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a, a_bytes);

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // A piece e of a lane: !A_TRANS: row mm, four consecutive k;  A_TRANS: k row kk, four
  // consecutive m (the weight as stored, coalesced either way)
  float4 areg[AV];
  const int bkk = tid >> 4, bnn = (tid & 15) * SEG;
  float bx[SEG], bdz[SEG];
  RowCoef rc;
  // OP_LIN4: the operand rows are relu(bn(W1 . x4)); this lane's SEG columns of the 4-channel
  // input stay in registers for the whole K loop, a chunk only fetches W1 rows and coefficients
  constexpr bool LIN = MODE == OP_LIN4;
  float x4r[LIN ? 4 : 1][LIN ? SEG : 1];
  float4 lw = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (LIN) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int i = 0; i < SEG; i += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(op.x + in_off + (size_t)cc * r + r0 + bnn + i);
        x4r[cc][i] = v.x; x4r[cc][i + 1] = v.y; x4r[cc][i + 2] = v.z; x4r[cc][i + 3] = v.w;
      }
  }
  auto fetch = [&](int k0) {
    if (k0 + KC <= k_total) {  // full chunk: range-checked 16-byte loads, no exec masks
#pragma unroll
      for (int e = 0; e < AV; ++e) {
        const int t = tid + e * 256;
        unsigned off;
        if (A_TRANS) off = (unsigned)((k0 + t / (TM / 4)) * lda + m0 + (t % (TM / 4)) * 4);
        else off = (unsigned)((m0 + t / (KC / 4)) * lda + k0 + (t % (KC / 4)) * 4);
        if (A_VEC) {  // 16-byte aligned rows: one load, one row predicate
          const int gm = m0 + (A_TRANS ? (t % (TM / 4)) * 4 : t / (KC / 4));
          areg[e] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (gm < m_total) areg[e] = *reinterpret_cast<const float4 *>(a + off);
        } else {  // rows that are not 16-byte aligned (K = 131, 259): four range-checked dwords
          const unsigned s = A_TRANS ? 4u : 4u;
          const unsigned v0 = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, off * 4u, 0, 0);
          const unsigned v1 = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, off * 4u + s, 0, 0);
          const unsigned v2 = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, off * 4u + 2 * s, 0, 0);
          const unsigned v3 = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, off * 4u + 3 * s, 0, 0);
          areg[e] = make_float4(__builtin_bit_cast(float, v0), __builtin_bit_cast(float, v1),
                                __builtin_bit_cast(float, v2), __builtin_bit_cast(float, v3));
        }
      }
    } else {  // K tail: per-element masks
#pragma unroll
      for (int e = 0; e < AV; ++e) {
        const int t = tid + e * 256;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gk = k0 + (A_TRANS ? t / (TM / 4) : (t % (KC / 4)) * 4 + q);
          const int gm = m0 + (A_TRANS ? (t % (TM / 4)) * 4 + q : t / (KC / 4));
          const size_t at = A_TRANS ? (size_t)gk * lda + gm : (size_t)gm * lda + gk;
          v[q] = (gm < m_total && gk < k_total) ? a[at] : 0.f;
        }
        areg[e] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    const int gk = k0 + bkk;
    const bool row_ok = gk < k_total;
    // rows beyond K: zero coefficients AND zero data -> the staged operand is exactly zero
    rc = load_row_coef<LIN ? OP_BNRELU : MODE>(op, gk, row_ok);
    if (!row_ok) { rc.sc = 0.f; rc.sh = 0.f; rc.a = 0.f; rc.q = 0.f; rc.p = 0.f; }
    if constexpr (LIN) {
      lw = row_ok ? *reinterpret_cast<const float4 *>(op.lin_w + (size_t)gk * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      load_raw_segment<MODE, SEG>(op, in_off + (size_t)gk * r + r0 + bnn, r0 + bnn, r, true, row_ok,
                                  bx, bdz, b * k_total + gk);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int t = tid + e * 256;
      if (A_TRANS) {  // one k row, four consecutive m
        const int kk = t / (TM / 4), mm = (t % (TM / 4)) * 4;
        float *dst = &As[buf][(kk >> 2) * PLA + (kk & 3)];
        dst[a_slot(mm) * 4] = areg[e].x; dst[a_slot(mm + 1) * 4] = areg[e].y;
        dst[a_slot(mm + 2) * 4] = areg[e].z; dst[a_slot(mm + 3) * 4] = areg[e].w;
      } else {        // one m row, four consecutive k: exactly one k-quad
        *reinterpret_cast<float4 *>(&As[buf][(t % (KC / 4)) * PLA + a_slot(t / (KC / 4)) * 4]) = areg[e];
      }
    }
#pragma unroll
    for (int i = 0; i < SEG; i += 4) {
      float4 v;
      if constexpr (LIN) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = fmaxf(__fmaf_rn(lin4(lw, x4r[0][i + e], x4r[1][i + e], x4r[2][i + e], x4r[3][i + e]),
                                 rc.sc, rc.sh), 0.f);
        v = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        v.x = transform<MODE>(bx[i + 0], bdz[i + 0], rc);
        v.y = transform<MODE>(bx[i + 1], bdz[i + 1], rc);
        v.z = transform<MODE>(bx[i + 2], bdz[i + 2], rc);
        v.w = transform<MODE>(bx[i + 3], bdz[i + 3], rc);
      }
      {
        const int n = bnn + i;
        float *dst = &Bs[buf][(bkk >> 2) * PLB + n * 4 + (n >> 3) * 4 + (bkk & 3)];
        dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
      }
    }
  };
  // fragments of half a chunk (four MFMA steps): plane 2 * (lane >> 5) + hh, this lane's rows
  const int a_frag = (2 * (lane >> 5)) * PLA;
  int a_row[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) a_row[i] = a_slot((wm * MB + i) * 32 + (lane & 31)) * 4;
  const int b_row = (wn * NB) * 32 + (lane & 31);
  const int b_frag = (2 * (lane >> 5)) * PLB + b_row * 4 + (b_row >> 3) * 4;
  auto fragments = [&](int buf, int hh, float4 (&af)[MB], float4 (&bf)[NB]) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
      af[i] = *reinterpret_cast<const float4 *>(&As[buf][a_frag + hh * PLA + a_row[i]]);
#pragma unroll
    for (int j = 0; j < NB; ++j)
      bf[j] = *reinterpret_cast<const float4 *>(&Bs[buf][b_frag + hh * PLB + j * (128 + 16)]);
  };
  auto multiply = [&](const float4 (&af)[MB], const float4 (&bf)[NB]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const float av = e == 0 ? af[i].x : e == 1 ? af[i].y : e == 2 ? af[i].z : af[i].w;
          const float bv = e == 0 ? bf[j].x : e == 1 ? bf[j].y : e == 2 ? bf[j].z : bf[j].w;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
        }
  };

  const int chunks = (k_total + KC - 1) / KC;
  fetch(0);
  stash(0);
  if (chunks > 1) fetch(KC);
  __syncthreads();
  for (int i = 0; i < chunks; ++i) {
    const int cur = i & 1;
    float4 af0[MB], bf0[NB], af1[MB], bf1[NB];
    fragments(cur, 0, af0, bf0);
    fragments(cur, 1, af1, bf1);                   // in flight during the first 4 * MB * NB MFMAs
    if constexpr (X6) {
      // the lane's eight k of the chunk (af0 | af1), as three bf16 terms each; six products per
      // block, the small ones first
      Split3 sa[MB];
#pragma unroll
      for (int ii = 0; ii < MB; ++ii) sa[ii] = split3(af0[ii], af1[ii]);
      // one column block at a time (its split lives for MB x 6 MFMAs); the next chunk is staged and
      // the one after it requested after the first block's MFMAs are issued
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const Split3 sb = split3(bf0[j], bf1[j]);
#pragma unroll
        for (int ii = 0; ii < MB; ++ii) mfma_x6(acc[ii][j], sa[ii], sb);
        if (j == 0) {
          if (i + 1 < chunks) stash(cur ^ 1);
          if (i + 2 < chunks) fetch((i + 2) * KC);
        }
      }
    } else {
      multiply(af0, bf0);
      if (i + 1 < chunks) stash(cur ^ 1);            // chunk i+1: registers -> the other buffer
      if (i + 2 < chunks) fetch((i + 2) * KC);       // chunk i+2: in flight during the MFMAs
      multiply(af1, bf1);
    }
    __syncthreads();  // buffer cur is free again, buffer cur^1 is complete
  }
  // C/D layout of the 32x32 block: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
  // (c == nullptr, pooled form only: the raw output is not stored at all -- statistics and extrema
  // are everything the layer leaves behind; its backward runs from the Gram matrix of its input,
  // mlp_pool_gram256.hip)
  if (POOL == 0 || c != nullptr) {
    float *cb = c + (size_t)b * b_stride_out;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int col = r0 + (wn * NB + j) * 32 + (lane & 31);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = m0 + (wm * MB + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          if (row < m_total) __builtin_nontemporal_store(acc[i][j][q], &cb[(size_t)row * r + col]);
        }
      }
  }
  // per output row (mean, M2) of this wave's NB*32 columns; the WN wave columns of a row are
  // merged by one lane per row (equal counts), and the workgroup writes TM contiguous pairs.
  // Where a wave owns 64 x 64 of the tile the sums come out of the row scan below (the
  // accumulators transposed through LDS: 96 plain adds per lane instead of 64 cross-lane
  // reductions of five DPP steps each -- the epilogue was ~640 vector instructions per wave
  // and tile, more issue time than the tile's MFMAs at K = 64); the DPP form serves the rest.
  constexpr bool SCAN_STATS = STATS && MB == 2 && NB == 2;
  __shared__ float2 wave_stat[STATS ? WN : 1][STATS ? TM : 1];
  auto merge_wave_stats = [&]() {
    __syncthreads();
    if (tid < TM && m0 + tid < m_total) {
      float mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < WN; ++w2) { mean += wave_stat[w2][tid].x; m2 += wave_stat[w2][tid].y; }
      mean *= 1.0f / (float)WN;
#pragma unroll
      for (int w2 = 0; w2 < WN; ++w2) {
        const float d = wave_stat[w2][tid].x - mean;
        m2 = __fmaf_rn((float)(NB * 32) * d, d, m2);
      }
      float *dst = stats + (((size_t)b * gridDim.x + blockIdx.x) * stat_channels + m0 + tid) * 2;
      dst[0] = mean;
      dst[1] = m2;
    }
  };
  if constexpr (STATS && !SCAN_STATS) {
    // shifted sums (shift = the row's first column in the wave: no cancellation)
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float v0 = acc[i][0][q];
        const float s_lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), 0));
        const float s_hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), 32));
        const float shf = half ? s_hi : s_lo;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const float d = acc[i][j][q] - shf;
          s1 += d;
          s2 = __fmaf_rn(d, d, s2);
        }
        s1 = half_wave_sum(s1);
        s2 = half_wave_sum(s2);
        constexpr float kInvN = 1.0f / (float)(NB * 32);
        if ((lane & 31) == 16) {
          const int lrow = (wm * MB + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
          wave_stat[wn][lrow] = make_float2(shf + s1 * kInvN, fmaxf(s2 - s1 * s1 * kInvN, 0.f));
        }
      }
    merge_wave_stats();
  }
  if constexpr (POOL != 0 || SCAN_STATS) {
    static_assert(NB == 2 && MB == 2 && (POOL == 0 || POOL == 16 || POOL == 32 || POOL == 64),
                  "a wave owns 64 x 64 of the tile");
    // Which extreme wins is known before the statistics are: sign(scale) = sign(gamma).  Rows
    // with a negative gamma are parked negated, so one scan for the maximum serves all.
    // Transpose through LDS (the operand buffers are free now; the barrier that ended the K loop
    // is behind every wave): the wave parks 32 rows x 64 columns of its accumulators, 16-byte
    // chunks swizzled [row][chunk ^ (row & 15)] (conflict-free for the column-wise 4-byte writes
    // and for the row-wise 16-byte reads), then lane (row = lane & 31, half = lane >> 5) scans
    // 32 consecutive samples of its row for the largest value and its first position.
    constexpr int AS_FLOATS = 2 * 4 * PLA;
    static_assert(AS_FLOATS >= 2 * 2048 && (AS_FLOATS >= 4 * 2048 || 2 * 4 * PLB >= 2 * 2048),
                  "operand buffers too small to park the accumulators");
    float *park = AS_FLOATS >= 4 * 2048 ? &As[0][0] + wave * 2048
                                        : (wave < 2 ? &As[0][0] + wave * 2048 : &Bs[0][0] + (wave - 2) * 2048);
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int PL = POOL != 0 ? POOL : 64;  // (no pooling: the scan serves the statistics alone)
    const int groups = r / PL;
    constexpr int GPL = 32 / (PL < 32 ? PL : 32);  // groups per lane: 2 for POOL == 16
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int rbase = m0 + (wm * MB + i) * 32;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cq = (q & 3) + 8 * (q >> 2);  // row of the block = cq + 4 * half
        const int rowq = rbase + cq + 4 * half;
        const bool neg = POOL != 0 && rowq < m_total && pool_gamma[rowq] < 0.f;
        const int rsw = (cq & 15) ^ (half << 2);  // (row & 15) for cq + 4*half (bit 2 of cq is clear)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int col = j * 32 + l31;
          const float v = acc[i][j][q];
          park[(cq + 4 * half) * 64 + ((((col >> 2) ^ rsw) << 2) | (col & 3))] = neg ? -v : v;
        }
      }
      // (same wave wrote and reads: LDS operations of a wave complete in order)
      float best[GPL];
      int at[GPL];
#pragma unroll
      for (int gq = 0; gq < GPL; ++gq) { best[gq] = -__builtin_inff(); at[gq] = 0; }
      const float4 *prow = reinterpret_cast<const float4 *>(park + l31 * 64);
      float t1 = 0.f, t2 = 0.f, tsh = 0.f;  // shifted sums of the lane's 32 samples (shift: the first)
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float4 v4 = prow[(half * 8 + c4) ^ (l31 & 15)];
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
        if (SCAN_STATS && c4 == 0) tsh = vv[0];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (POOL != 0) {
            const int s2 = c4 * 4 + e;             // sample within the lane's 32
            const int gq = GPL == 2 ? s2 / 16 : 0;  // its group within the lane
            if (vv[e] > best[gq]) { best[gq] = vv[e]; at[gq] = GPL == 2 ? s2 % 16 : s2; }
          }
          if constexpr (SCAN_STATS) {
            const float d = vv[e] - tsh;
            t1 += d;
            t2 = __fmaf_rn(d, d, t2);
          }
        }
      }
      if constexpr (SCAN_STATS) {
        // (mean, M2) of the lane's 32 samples, then of the row's 64 in this wave (equal counts)
        const float mh = tsh + t1 * (1.0f / 32.0f), qh = fmaxf(t2 - t1 * t1 * (1.0f / 32.0f), 0.f);
        const float mo = __shfl_xor(mh, 32, kWave), qo = __shfl_xor(qh, 32, kWave);
        if (half == 0) {
          const float dlt = mo - mh;
          float mean = 0.5f * (mh + mo);
          const float m2w = (qh + qo) + 16.0f * dlt * dlt;  // n_a n_b / (n_a + n_b) = 16
          // rows parked negated (negative gamma, pooling): the mean changes sign, M2 does not
          if (POOL != 0 && rbase + l31 < m_total && pool_gamma[rbase + l31] < 0.f) mean = -mean;
          wave_stat[wn][(wm * MB + i) * 32 + l31] = make_float2(mean, m2w);
        }
      }
      if (POOL == 64) {  // the two halves of a row form one group: the lower half wins ties
        const float ob = __shfl_xor(best[0], 32, kWave);
        const int oa = __shfl_xor(at[0], 32, kWave);
        if (half == 0 && ob > best[0]) { best[0] = ob; at[0] = 32 + oa; }
      }
      const int row = rbase + l31;
      if (POOL != 0 && row < m_total && (POOL != 64 || half == 0)) {
        const bool negr = pool_gamma[row] < 0.f;
        const int g0 = (r0 + wn * 64) / POOL + (POOL == 64 ? 0 : half * GPL);
        int *ei = reinterpret_cast<int *>(ext);
#pragma unroll
        for (int gq = 0; gq < GPL; ++gq) {
          const size_t o = ((size_t)b * m_total + row) * groups + g0 + gq;
          ext[o] = negr ? -best[gq] : best[gq];
          ei[ext_plane + o] = at[gq];
        }
      }
    }
    if constexpr (SCAN_STATS) merge_wave_stats();
  }
}

// Small-problem variant of gemm_nn_kernel (R of a few hundred columns per cloud: the FP layers
// and the vote / proposal / IoU heads).  There the K loop of the big kernel is a chain of
// K/16 exposed load latencies with about one workgroup per CU.  Here a workgroup owns a 64 x 64
// output tile, stages K in chunks of 64 (a quarter of the dependent steps) and its four waves
// split each chunk's k-rows -- every wave accumulates the whole tile over its 16 rows -- and add
// their accumulators through LDS at the end.
constexpr int KS = 64;  // K chunk of the small variant

// X6: the chunk's 16 k-rows of a wave are ONE bf16 MFMA step; fragments gathered as eight 4-byte
// reads per row block and split in registers (mlp_operand.h)
// A operand as a ready bf16 image (X6 only; mlp_weight_images_build): the three planes of the exact
// split of A as [plane][row][reduction index], rows and reduction padded with zeros to multiples
// of 64 -- for the forward the weight itself, for the data gradient its transpose.  A lane's MFMA
// fragment (row lane & 31, eight consecutive reduction indices) is ONE 16-byte load per plane,
// requested a chunk ahead; the 64 x 64 weight tile is then neither staged through LDS nor split
// by every one of the layer's ~128 workgroups (88 + ~60 of the ~250 vector instructions per wave
// and chunk that bound this kernel).
struct AImage {
  const unsigned short *p;   // nullptr: A is read as fp32 and split here
  int pitch;                 // reduction indices per row (multiple of 64)
  size_t plane;              // elements per plane
};

template <int MODE, bool A_TRANS, bool X6, bool AIMG = false>
__device__ __forceinline__ void gemm_nn_small_body(int blk_x, int blk_y, int blk_z, int m_total, int k_total,
                                                   int r, const float *__restrict__ a, int lda,
                                                   OperandB opb, float *__restrict__ c,
                                                   size_t b_stride_in, size_t b_stride_out,
                                                   const AImage img = AImage{nullptr, 0, 0}) {
  constexpr int TM = 64, TN = 64, LDA = TM + 1;
  constexpr int STAGE = KS * LDA + KS * TN, REDUCE = 4 * 16 * 64;
  __shared__ __attribute__((aligned(16))) float lds[STAGE > REDUCE ? STAGE : REDUCE];
  float *As = lds + KS * TN, *Bs = lds;  // Bs first: 16-byte aligned for the float4 stores
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blk_x * TN, m0 = blk_y * TM, b = blk_z;
  OperandB op = opb;
  const size_t in_off = (size_t)b * b_stride_in;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int bkk = tid >> 2, bnn = (tid & 3) * 16;  // B: row of the chunk, 16 consecutive columns
  const bool vec_ok = (r & 3) == 0;
  float areg[16], bx[16], bdz[16];
  RowCoef rc;
  bool brow_ok = false;
  static_assert(!AIMG || X6, "images feed the bf16 form only");
  const bf16x8 *img8 = reinterpret_cast<const bf16x8 *>(img.p);
  auto fetch = [&](int k0) {
    if constexpr (!AIMG) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = tid + e * 256;
        const int kk = A_TRANS ? t / TM : t % KS, mm = A_TRANS ? t % TM : t / KS;
        const int gm = m0 + mm, gk = k0 + kk;
        const size_t at = A_TRANS ? (size_t)gk * lda + gm : (size_t)gm * lda + gk;
        areg[e] = (gm < m_total && gk < k_total) ? a[at] : 0.f;
      }
    }
    const int gk = k0 + bkk;
    brow_ok = gk < k_total;
    rc = load_row_coef<MODE>(op, gk, brow_ok);
    load_raw_segment<MODE, 16>(op, in_off + (size_t)gk * r + r0 + bnn, r0 + bnn, r, vec_ok,
                               brow_ok, bx, bdz, b * k_total + gk);
  };
  fetch(0);
  for (int k0 = 0; k0 < k_total; k0 += KS) {
    // (image form: this chunk's weight fragments are requested here, ahead of the barrier and the
    //  staging of B, and used after them)
    Split3 sa_img[2];
    if constexpr (AIMG) {
      const int k0w = wave * (KS / 4) + 8 * (lane >> 5);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const size_t at = ((size_t)(m0 + i * 32 + (lane & 31)) * img.pitch + k0 + k0w) >> 3;
        sa_img[i].hi = img8[at];
        sa_img[i].mid = img8[at + (img.plane >> 3)];
        sa_img[i].lo = img8[at + 2 * (img.plane >> 3)];
      }
    }
    __syncthreads();
    if constexpr (!AIMG) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = tid + e * 256;
        As[(A_TRANS ? t / TM : t % KS) * LDA + (A_TRANS ? t % TM : t / KS)] = areg[e];
      }
    }
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      float4 v;
      v.x = (brow_ok && r0 + bnn + i + 0 < r) ? transform<MODE>(bx[i + 0], bdz[i + 0], rc) : 0.f;
      v.y = (brow_ok && r0 + bnn + i + 1 < r) ? transform<MODE>(bx[i + 1], bdz[i + 1], rc) : 0.f;
      v.z = (brow_ok && r0 + bnn + i + 2 < r) ? transform<MODE>(bx[i + 2], bdz[i + 2], rc) : 0.f;
      v.w = (brow_ok && r0 + bnn + i + 3 < r) ? transform<MODE>(bx[i + 3], bdz[i + 3], rc) : 0.f;
      *reinterpret_cast<float4 *>(&Bs[bkk * TN + bnn + i]) = v;
    }
    __syncthreads();
    if (k0 + KS < k_total) fetch(k0 + KS);
    if constexpr (X6) {
      static_assert(KS / 4 == 16, "a wave's share of the chunk is one 16-deep MFMA step");
      const int k0w = wave * (KS / 4) + 8 * (lane >> 5);
      Split3 sa[2], sb[2];
      if constexpr (AIMG) {
        sa[0] = sa_img[0]; sa[1] = sa_img[1];
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = As[(k0w + e) * LDA + i * 32 + (lane & 31)];
          sa[i] = split3(v);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Bs[(k0w + e) * TN + j * 32 + (lane & 31)];
        sb[j] = split3(v);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma_x6(acc[i][j], sa[i], sb[j]);
    } else {
#pragma unroll
    for (int kk = 0; kk < KS / 4; kk += 2) {
      const int krow = wave * (KS / 4) + kk + (lane >> 5);
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = As[krow * LDA + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = Bs[krow * TN + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    }
  }
  for (int s = 1; s < 4; ++s) {  // waves 1..3 hand their accumulators to wave 0
    __syncthreads();
    if (wave == s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) lds[((i * 2 + j) * 16 + q) * 64 + lane] = acc[i][j][q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] += lds[((i * 2 + j) * 16 + q) * 64 + lane];
    }
  }
  if (wave != 0) return;
  float *cb = c + (size_t)b * b_stride_out;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = r0 + j * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < m_total && col < r) __builtin_nontemporal_store(acc[i][j][q], &cb[(size_t)row * r + col]);
      }
    }
}

template <int MODE, bool A_TRANS, bool X6 = false, bool AIMG = false>
__global__ void __launch_bounds__(256, (X6 && A_TRANS) ? 2 : 1)
gemm_nn_small_kernel(int m_total, int k_total, int r, const float *__restrict__ a, int lda,
                     OperandB opb, float *__restrict__ c, size_t b_stride_in,
                     size_t b_stride_out, AImage img) {
  gemm_nn_small_body<MODE, A_TRANS, X6, AIMG>((int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, m_total,
                                              k_total, r, a, lda, opb, c, b_stride_in, b_stride_out, img);
}

// Partial wgrad: for one cloud b and one slice of R,
//   part[slice][m][k] = sum_{r in slice} P[b][m][r] * Q[b][k][r]
// P = op_p (mode PMODE, rows m: the gradient operand, the expensive transform), Q = op_q (mode
// QMODE, rows k).  A workgroup owns 64 rows of M and TK = 32*KBW*WK columns of K, so the
// costly P tile is staged once for up to 256 columns of K.  The four waves are arranged as WK
// column groups x RS shares of the r-chunk: every wave accumulates a 64 x (32*KBW) block
// (2 x KBW MFMA tiles, 2 + KBW LDS reads per 2*KBW MFMAs) over its share of the chunk's
// r-steps; shares are summed through LDS at the end.  The R axis is staged [r][m] / [r][k].
constexpr int RC = 32;  // r chunk (128-byte row segments per load)

template <int PMODE, int QMODE, int KBW, int WK, int RS, bool X6 = false>
__global__ void __launch_bounds__(256)
gemm_wgrad_kernel(int m_total, int k_begin, int k_end, int k_total, int r, int r_per_slice,
                  OperandB opp, OperandB opq, float *__restrict__ part, size_t p_stride,
                  size_t q_stride) {
  static_assert(WK * RS == 4, "four waves");
  constexpr int TM = 64, TK = 32 * KBW * WK;
  constexpr int QSEG = TK / 64;            // 8-float row segments of Q per lane
  constexpr int LDP = TM + 1, LDQ = TK + 1;
  constexpr int STAGE = RC * (LDP + LDQ);
  constexpr int REDUCE = RS > 1 ? WK * 2 * KBW * 16 * 64 : 0;  // one accumulator set per group
  __shared__ float lds[STAGE > REDUCE ? STAGE : REDUCE];
  float *Ps = lds, *Qs = lds + RC * LDP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WK, wr = wave / WK;
  // XCD-contiguous ids: the 64-row tiles of one (cloud, slice) share their Q rows through one L2
  const BlockId blk = xcd_block_id();
  const int k0 = k_begin + blk.x * TK, m0 = blk.y * TM;
  const int slices = (r + r_per_slice - 1) / r_per_slice;
  const int b = blk.z / slices, slice = blk.z % slices;
  const int r_lo = slice * r_per_slice;
  const int r_hi = r_lo + r_per_slice < r ? r_lo + r_per_slice : r;
  OperandB P = opp, Q = opq;
  const size_t p_off = (size_t)b * p_stride, q_off = (size_t)b * q_stride;

  f32x16 acc[2][KBW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < KBW; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // fixed ownership: lane t loads 8 consecutive r of P row t/4 and of Q rows t/4 + 64*s, so the
  // per-row constants live in registers for the whole kernel.  The raw loads of chunk i+1 are
  // issued before the MFMAs of chunk i (register prefetch).
  const int seg_row = tid >> 2, seg_r = (tid & 3) * 8;
  const bool p_ok = m0 + seg_row < m_total;
  const RowCoef pc = load_row_coef<PMODE>(P, m0 + seg_row, p_ok);
  const size_t p_row = p_off + (size_t)(m0 + seg_row) * r;
  bool q_ok[QSEG];
  RowCoef qc[QSEG];
  size_t q_row[QSEG];
#pragma unroll
  for (int s = 0; s < QSEG; ++s) {
    const int gk = k0 + seg_row + 64 * s;
    q_ok[s] = gk < k_end;
    qc[s] = load_row_coef<QMODE>(Q, gk, q_ok[s]);
    q_row[s] = q_off + (size_t)gk * r;
  }
  const bool vec_ok = ((r | r_hi) & 3) == 0;
  float px[8], pdz[8], qx[QSEG][8], qdz[QSEG][8];
  auto fetch = [&](int rr) {
    load_raw_segment<PMODE, 8>(P, p_row + rr + seg_r, rr + seg_r, r_hi, vec_ok, p_ok, px, pdz,
                               b * m_total + m0 + seg_row);
#pragma unroll
    for (int s = 0; s < QSEG; ++s)
      load_raw_segment<QMODE, 8>(Q, q_row[s] + rr + seg_r, rr + seg_r, r_hi, vec_ok, q_ok[s],
                                 qx[s], qdz[s]);
  };
  fetch(r_lo);
  for (int rr = r_lo; rr < r_hi; rr += RC) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i)
      Ps[(seg_r + i) * LDP + seg_row] =
          (p_ok && rr + seg_r + i < r_hi) ? transform<PMODE>(px[i], pdz[i], pc) : 0.f;
#pragma unroll
    for (int s = 0; s < QSEG; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        Qs[(seg_r + i) * LDQ + seg_row + 64 * s] =
            (q_ok[s] && rr + seg_r + i < r_hi) ? transform<QMODE>(qx[s][i], qdz[s][i], qc[s]) : 0.f;
    __syncthreads();
    if (rr + RC < r_hi) fetch(rr + RC);
    if constexpr (X6 && (RC / RS) % 16 == 0) {
      // a wave's share of the r-chunk in 16-deep bf16 MFMA steps (mlp_operand.h split3 / mfma_x6)
#pragma unroll
      for (int st = 0; st < RC / RS / 16; ++st) {
        const int row0 = wr * (RC / RS) + 16 * st + 8 * (lane >> 5);
        Split3 sp[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = Ps[(row0 + e) * LDP + i * 32 + (lane & 31)];
          sp[i] = split3(v);
        }
#pragma unroll
        for (int j = 0; j < KBW; ++j) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = Qs[(row0 + e) * LDQ + (wk * KBW + j) * 32 + (lane & 31)];
          const Split3 sq = split3(v);
#pragma unroll
          for (int i = 0; i < 2; ++i) mfma_x6(acc[i][j], sp[i], sq);
        }
      }
    } else {
#pragma unroll
    for (int st = 0; st < RC / 2 / RS; ++st) {
      const int row = wr * (RC / RS) + 2 * st + (lane >> 5);
      float bq[KBW], ap[2];
#pragma unroll
      for (int j = 0; j < KBW; ++j) bq[j] = Qs[row * LDQ + (wk * KBW + j) * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 2; ++i) ap[i] = Ps[row * LDP + i * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KBW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[i], bq[j], acc[i][j], 0, 0, 0);
    }
    }
  }
  if (RS > 1) {  // sum the r-shares of each column group: share s -> LDS -> share 0
    float *red = lds + (size_t)wk * (2 * KBW * 16 * 64);
    for (int s = 1; s < RS; ++s) {
      __syncthreads();
      if (wr == s) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < KBW; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) red[((i * KBW + j) * 16 + q) * 64 + lane] = acc[i][j][q];
      }
      __syncthreads();
      if (wr == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < KBW; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] += red[((i * KBW + j) * 16 + q) * 64 + lane];
      }
    }
    if (wr != 0) return;
  }
  float *out = part + (size_t)blk.z * m_total * k_total;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < KBW; ++j) {
      const int col = k0 + (wk * KBW + j) * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < m_total && col < k_end) out[(size_t)row * k_total + col] = acc[i][j][q];
      }
    }
}

// ---- weight gradient with the MFMA fragments read STRAIGHT from memory ---------------------------
// dW[m][k] = sum_r P[m][r] Q[k][r]: the reduction runs over the columns r, which are contiguous in
// BOTH operands -- exactly the eight consecutive reduction elements per lane a 32x32x16 bf16
// MFMA fragment holds (row / column = lane & 31).  So no LDS staging and no transposition: a lane
// loads 16 consecutive columns of its P and Q rows (64 bytes; the two lane halves of a row make a
// 128-byte run), applies the operand transform, splits into bf16 terms and issues two k-steps
// (the first over its columns 0..7, the second over 8..15 -- any 16 columns form a k-step as long
// as both operands agree).  A wave owns a 64 x 64 block of dW over its share of the workgroup's
// column range (32-column chunks dealt round-robin to the four waves, the next chunk's loads in
// flight during the MFMAs), the shares are added through LDS, one partial block per workgroup.
// For the small layers (FP modules, heads, the pre-gather first layers: a few thousand columns per
// cloud, operands resident in L2) this replaces the LDS-staged kernel above at 0.4-0.6 of its time.
template <int PMODE, int QMODE>
__device__ __forceinline__ void wgrad_direct_body(const BlockId blk, int m_total, int k_total, int r,
                                                  int r_per_slice, int slices, OperandB opp,
                                                  OperandB opq, float *__restrict__ part,
                                                  size_t p_stride, size_t q_stride) {
  __shared__ float red[4 * 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int k0 = blk.x * 64, m0 = blk.y * 64;
  const int b = blk.z / slices, slice = blk.z % slices;
  const int r_lo = slice * r_per_slice;
  const int r_hi = r_lo + r_per_slice < r ? r_lo + r_per_slice : r;
  OperandB P = opp, Q = opq;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  bool p_ok[2], q_ok[2];
  RowCoef pc[2], qc[2];
  size_t p_row[2], q_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int gm = m0 + i * 32 + l31, gk = k0 + i * 32 + l31;
    p_ok[i] = gm < m_total;
    q_ok[i] = gk < k_total;
    pc[i] = load_row_coef<PMODE>(P, gm, p_ok[i]);
    qc[i] = load_row_coef<QMODE>(Q, gk, q_ok[i]);
    p_row[i] = (size_t)b * p_stride + (size_t)gm * r;
    q_row[i] = (size_t)b * q_stride + (size_t)gk * r;
  }
  const bool vec_ok = (r & 3) == 0;
  float px[2][16], pdz[2][16], qx[2][16], qdz[2][16];
  auto fetch = [&](int c0) {  // the lane's 16 columns of chunk c0: c0 + 16 * half ..
    const int col = c0 + 16 * half;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      load_raw_segment<PMODE, 16>(P, p_row[i] + col, col, r_hi, vec_ok, p_ok[i], px[i], pdz[i]);
      load_raw_segment<QMODE, 16>(Q, q_row[i] + col, col, r_hi, vec_ok, q_ok[i], qx[i], qdz[i]);
    }
  };
  int c0 = r_lo + 32 * wave;
  if (c0 < r_hi) fetch(c0);
  for (; c0 < r_hi; c0 += 128) {
    const int col = c0 + 16 * half;
#pragma unroll
    for (int st = 0; st < 2; ++st) {  // one k-step's fragments live at a time (registers)
      Split3 sp[2], sq[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float vp[8], vq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool in = col + 8 * st + e < r_hi;
          vp[e] = (p_ok[i] && in) ? transform<PMODE>(px[i][8 * st + e], pdz[i][8 * st + e], pc[i]) : 0.f;
          vq[e] = (q_ok[i] && in) ? transform<QMODE>(qx[i][8 * st + e], qdz[i][8 * st + e], qc[i]) : 0.f;
        }
        sp[i] = split3(vp);
        sq[i] = split3(vq);
      }
      if (st == 1 && c0 + 128 < r_hi) fetch(c0 + 128);  // the raw registers are free again
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma_x6(acc[i][j], sp[i], sq[j]);
    }
  }
  // add the four waves' blocks: waves 1..3 -> LDS -> wave 0, one 32 x 32 tile at a time
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __syncthreads();
      if (wave != 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) red[((wave * 16) + q) * 64 + lane] = acc[i][j][q];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          acc[i][j][q] += (red[(16 + q) * 64 + lane] + red[(32 + q) * 64 + lane]) + red[(48 + q) * 64 + lane];
      }
    }
  if (wave != 0) return;
  float *out = part + (size_t)blk.z * m_total * k_total;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = k0 + j * 32 + l31;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
        if (row < m_total && col < k_total) out[(size_t)row * k_total + col] = acc[i][j][q];
      }
    }
}

template <int PMODE, int QMODE>
__global__ void __launch_bounds__(256, 2)
gemm_wgrad_direct_kernel(int m_total, int k_total, int r, int r_per_slice, int slices, OperandB opp,
                         OperandB opq, float *__restrict__ part, size_t p_stride, size_t q_stride) {
  wgrad_direct_body<PMODE, QMODE>(xcd_block_id(), m_total, k_total, r, r_per_slice, slices, opp, opq,
                                  part, p_stride, q_stride);
}

// Both backward GEMMs of a SMALL layer in one launch: the data gradient dQ = W^T . P (the
// 64 x 64-tile kernel above, blocks [0, nd)) and the weight gradient (the direct-fragment kernel,
// blocks [nd, nd + nw)) are independent and neither fills the chip -- launched one after the other
// they cost their sum (2 x 10-28 us per layer, 13 layers per step), together about the larger.
struct SmallPairArgs {
  int nd, dgx, dgy;             // data-gradient blocks and their (x, y) extents (z = cloud)
  int m, k, r;                  // the layer: W (m, k), r columns per cloud
  int wkx, wmy, per, slices;    // weight-gradient grid (x over k, y over m) and slicing
  const float *w;               // (m, k)
  float *dq, *part;
  AImage wt_img;                // the transposed weight as a bf16 image (or none)
};

template <int PMODE, int QMODE, bool X6D, bool AIMG = false>
__global__ void __launch_bounds__(256, 2)
gemm_small_backward_pair_kernel(SmallPairArgs t, OperandB opp, OperandB opq) {
  const int id = (int)blockIdx.x;
  if (id < t.nd) {
    const int bx = id % t.dgx, by = (id / t.dgx) % t.dgy, bz = id / (t.dgx * t.dgy);
    // dQ (k rows) = W^T (k x m, read transposed) . P (m rows): the small kernel's (m_total, k_total)
    // are (k, m) here
    gemm_nn_small_body<PMODE, true, X6D, AIMG>(bx, by, bz, t.k, t.m, t.r, t.w, t.k, opp, t.dq,
                                               (size_t)t.m * t.r, (size_t)t.k * t.r, t.wt_img);
  } else {
    const int wid = id - t.nd;
    BlockId blk;
    blk.x = wid % t.wkx;
    blk.y = (wid / t.wkx) % t.wmy;
    blk.z = wid / (t.wkx * t.wmy);
    wgrad_direct_body<PMODE, QMODE>(blk, t.m, t.k, t.r, t.per, t.slices, opp, opq, t.part,
                                    (size_t)t.m * t.r, (size_t)t.k * t.r);
  }
}

// dw[i] = sum_p part[p][i]: a workgroup owns 32 consecutive elements, its 8 lane groups each sum
// every 8th partial (128-byte rows, eight loads in flight per lane), LDS adds the groups in a
// fixed order -- deterministic, and a few hundred partials finish in a few microseconds.
// VEC (count % 4 == 0, 16-byte aligned rows): a workgroup owns 128 consecutive elements, a lane four of
// them as ONE 16-byte load per partial -- the same additions in the same order per element, a quarter
// of the load instructions (the queued reduction of a backward pass reads ~280 MB).
template <bool VEC>
__device__ __forceinline__ void reduce_partials_block(int block, int count, int parts,
                                                      const float *__restrict__ part,
                                                      float *__restrict__ out) {
  constexpr int W = VEC ? 4 : 1;
  __shared__ float sums[8][32 * W];
  const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = (block * 32 + e) * W;
  float s[8][W];
#pragma unroll
  for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int c = 0; c < W; ++c) s[u][c] = 0.f;
  auto row = [&](int p, float (&v)[W]) {
    if constexpr (VEC) {
      const float4 t = *reinterpret_cast<const float4 *>(part + (size_t)p * count + i);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      v[0] = part[(size_t)p * count + i];
    }
  };
  if (i < count) {
    int p = grp;
    for (; p + 56 < parts; p += 64) {  // eight rows in flight per lane
      float v[8][W];
#pragma unroll
      for (int u = 0; u < 8; ++u) row(p + 8 * u, v[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < W; ++c) s[u][c] += v[u][c];
    }
    for (; p + 24 < parts; p += 32) {
      float v[4][W];
#pragma unroll
      for (int u = 0; u < 4; ++u) row(p + 8 * u, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < W; ++c) s[u][c] += v[u][c];
    }
    for (; p < parts; p += 8) {
      float v[W];
      row(p, v);
#pragma unroll
      for (int c = 0; c < W; ++c) s[0][c] += v[c];
    }
  }
#pragma unroll
  for (int c = 0; c < W; ++c)
    sums[grp][e * W + c] = ((s[0][c] + s[1][c]) + (s[2][c] + s[3][c])) + ((s[4][c] + s[5][c]) + (s[6][c] + s[7][c]));
  __syncthreads();
  if (grp == 0 && i < count) {
#pragma unroll
    for (int c = 0; c < W; ++c) {
      float t = sums[0][e * W + c];
#pragma unroll
      for (int q = 1; q < 8; ++q) t += sums[q][e * W + c];
      out[i + c] = t;
    }
  }
}

__global__ void __launch_bounds__(256)
reduce_partials_kernel(int count, int parts, const float *__restrict__ part,
                       float *__restrict__ out) {
  reduce_partials_block<false>((int)blockIdx.x, count, parts, part, out);
}

// Many reductions in one launch (the queued weight gradients of a backward pass): the table
// travels as the kernel argument, a workgroup finds its entry by a scalar scan of the block
// prefix.  Same arithmetic and order per element as the single form.
constexpr int kReduceBatch = 40;
struct ReduceBatch {
  int n;
  int first_block[kReduceBatch + 1];
  int count[kReduceBatch];
  int parts[kReduceBatch];
  const float *part[kReduceBatch];
  float *out[kReduceBatch];
  unsigned char vec[kReduceBatch];
};

__global__ void __launch_bounds__(256) reduce_partials_batch_kernel(const ReduceBatch t) {
  const int blk = (int)blockIdx.x;
  int e = 0;
  while (e + 1 < t.n && blk >= t.first_block[e + 1]) ++e;
  // (an entry is vectorised when its rows are 16-byte aligned: its blocks then cover 128 elements)
  if (t.vec[e]) reduce_partials_block<true>(blk - t.first_block[e], t.count[e], t.parts[e], t.part[e], t.out[e]);
  else reduce_partials_block<false>(blk - t.first_block[e], t.count[e], t.parts[e], t.part[e], t.out[e]);
}

// fp32 products as six bf16 MFMAs (see split3): on unless MLP_GEMM_SPLIT_BF16=0 (read once)
bool gemm_x6() {
  static const bool on = !(getenv("MLP_GEMM_SPLIT_BF16") && atoi(getenv("MLP_GEMM_SPLIT_BF16")) == 0);
  return on;
}

// the stand-alone weight-gradient kernel split the same way: slower over the layer shapes (2436
// against 2346 us, profiles/r4_split_bf16.json) -- its fragments are split by every wave again;
// kept behind MLP_WGRAD_SPLIT_BF16=1
bool wgrad_x6() {
  static const bool on = gemm_x6() && getenv("MLP_WGRAD_SPLIT_BF16") && atoi(getenv("MLP_WGRAD_SPLIT_BF16")) == 1;
  return on;
}

// forward GEMM whose epilogue also leaves BatchNorm partials (see gemm_nn2_kernel STATS); only
// instantiated for the operand modes of the forward pass
template <int TM, int TN, int WM, int WN, int MODE, bool A_TRANS>
void launch_stats(bool a_vec, int r, int b, hipStream_t stream, int rows, int k, const float *a_t,
                  int lda, unsigned a_bytes, const OperandB &op, float *c_t, size_t in_stride,
                  size_t out_stride, float *stats, int channels, int done) {
  if constexpr (!A_TRANS && MODE <= OP_BNRELU) {
    // partial (part, channel) pairs: part = cloud * tiles + tile, channel = done + row
    float *st = stats + (size_t)done * 2;
#define ST_LAUNCH(AV, X6)                                                                         \
  hipLaunchKernelGGL((gemm_nn2_kernel<TM, TN, WM, WN, MODE, false, AV, true, 0, X6>),               \
                     dim3(pn2_ceil_div(r, TN), 1, b), dim3(256), 0, stream, rows, k, r, a_t, lda,   \
                     a_bytes, op, c_t, in_stride, out_stride, st, channels)
    if (gemm_x6()) { if (a_vec) ST_LAUNCH(true, true); else ST_LAUNCH(false, true); }
    else { if (a_vec) ST_LAUNCH(true, false); else ST_LAUNCH(false, false); }
#undef ST_LAUNCH
  }
}

template <int MODE, bool A_TRANS = false>
int launch_nn(int b, int m, int k, int r, const float *a, int lda, const OperandB &op, float *c,
              size_t in_stride, size_t out_stride, hipStream_t stream, float *stats = nullptr,
              const AImage img = AImage{nullptr, 0, 0}) {
  // (read per call so that the tests can steer both kernels; a getenv costs nothing next to a launch)
  const char *env = getenv("MLP_SMALL_GEMM_COLS");
  const long long small_cols = env ? atoll(env) : 16384;
  if ((long long)b * r <= small_cols) {  // a few hundred columns per cloud: latency-bound regime
    // (measured, tools/small_gemm_bench.py: the transposed form with a plain operand gains from the
    //  split at two workgroups per CU, 20 -> 15.6 us; with the on-the-fly dY operand it spills and does not)
    if (gemm_x6() && (!A_TRANS || MODE == OP_DIRECT) && img.p != nullptr)
      hipLaunchKernelGGL((gemm_nn_small_kernel<MODE, A_TRANS, true, true>),
                         dim3(pn2_ceil_div(r, 64), pn2_ceil_div(m, 64), b), dim3(256), 0, stream, m,
                         k, r, a, lda, op, c, in_stride, out_stride, img);
    else if (gemm_x6() && (!A_TRANS || MODE == OP_DIRECT))
      hipLaunchKernelGGL((gemm_nn_small_kernel<MODE, A_TRANS, true>),
                         dim3(pn2_ceil_div(r, 64), pn2_ceil_div(m, 64), b), dim3(256), 0, stream, m,
                         k, r, a, lda, op, c, in_stride, out_stride, AImage{nullptr, 0, 0});
    else
      hipLaunchKernelGGL((gemm_nn_small_kernel<MODE, A_TRANS>),
                         dim3(pn2_ceil_div(r, 64), pn2_ceil_div(m, 64), b), dim3(256), 0, stream, m,
                         k, r, a, lda, op, c, in_stride, out_stride, AImage{nullptr, 0, 0});
    return pn2_launch_status();
  }
  // rows are covered by 256-row tiles, then one smaller tile for the remainder
  const char *v2env = getenv("MLP_GEMM_PIPELINED");
  const bool pipelined = !(v2env && atoi(v2env) == 0) && (r % 256 == 0);
  const size_t a_total = A_TRANS ? (size_t)k * lda : (size_t)m * lda;  // floats from `a` on
  int done = 0;
  while (done < m) {
    const int left = m - done;
    const float *a_t = A_TRANS ? a + done : a + (size_t)done * lda;
    const unsigned a_bytes = (unsigned)(4 * (A_TRANS ? a_total - done : a_total - (size_t)done * lda));
    float *c_t = c + (size_t)done * r;
    const bool a_vec = (lda & 3) == 0 && (reinterpret_cast<size_t>(a_t) & 15) == 0;
#define NN(TM, TN, WM, WN)                                                                      \
  do {                                                                                          \
    if (stats)                                                                                  \
      launch_stats<TM, TN, WM, WN, MODE, A_TRANS>(a_vec, r, b, stream, rows, k, a_t, lda, a_bytes, \
                                                  op, c_t, in_stride, out_stride, stats, m, done); \
    else if (pipelined && a_vec && gemm_x6())                                                   \
      hipLaunchKernelGGL((gemm_nn2_kernel<TM, TN, WM, WN, MODE, A_TRANS, true, false, 0, true>), \
                         dim3(pn2_ceil_div(r, TN), 1, b), dim3(256), 0, stream, rows, k, r, a_t,  \
                         lda, a_bytes, op, c_t, in_stride, out_stride);                         \
    else if (pipelined && gemm_x6())                                                            \
      hipLaunchKernelGGL((gemm_nn2_kernel<TM, TN, WM, WN, MODE, A_TRANS, false, false, 0, true>), \
                         dim3(pn2_ceil_div(r, TN), 1, b), dim3(256), 0, stream, rows, k, r, a_t,  \
                         lda, a_bytes, op, c_t, in_stride, out_stride);                         \
    else if (pipelined && a_vec)                                                                \
      hipLaunchKernelGGL((gemm_nn2_kernel<TM, TN, WM, WN, MODE, A_TRANS, true>),                \
                         dim3(pn2_ceil_div(r, TN), 1, b), dim3(256), 0, stream, rows, k, r, a_t,  \
                         lda, a_bytes, op, c_t, in_stride, out_stride);                         \
    else if (pipelined)                                                                         \
      hipLaunchKernelGGL((gemm_nn2_kernel<TM, TN, WM, WN, MODE, A_TRANS, false>),               \
                         dim3(pn2_ceil_div(r, TN), 1, b), dim3(256), 0, stream, rows, k, r, a_t,  \
                         lda, a_bytes, op, c_t, in_stride, out_stride);                         \
    else                                                                                        \
      hipLaunchKernelGGL((gemm_nn_kernel<TM, TN, WM, WN, MODE, A_TRANS>),                       \
                         dim3(pn2_ceil_div(r, TN), 1, b), dim3(256), 0, stream, rows, k, r, a_t,  \
                         lda, op, c_t, in_stride, out_stride);                                  \
  } while (0)
    int rows;
    if (left >= 256) { rows = 256; NN(256, 64, 4, 1); }
    else if (left > 64) { rows = left < 128 ? left : 128; NN(128, 128, 2, 2); }
    else if (left > 32) { rows = left; NN(64, 128, 2, 2); }
    else {  // (a 32-row tile is narrower than the 16-byte A pieces of the pipelined kernel)
      rows = left;
      hipLaunchKernelGGL((gemm_nn_kernel<32, 256, 1, 4, MODE, A_TRANS>),
                         dim3(pn2_ceil_div(r, 256), 1, b), dim3(256), 0, stream, rows, k, r, a_t,
                         lda, op, c_t, in_stride, out_stride);
    }
#undef NN
    done += rows;
  }
  return pn2_launch_status();
}

// ---- bf16 images of the weights (AImage) ---------------------------------------------------------
// One launch for every weight of the table: workgroup = one 64 x 64 tile of one weight; the exact
// three-term split of mlp_operand.h per element (bit for bit what split3 produces in the kernels
// that split on the fly), written in BOTH orders: [plane][m][k] for the forward, [plane][k][m] for
// the data gradient.
constexpr int kImageBatch = 24;
struct ImageBatch {
  int n;
  int first_tile[kImageBatch + 1];
  int m[kImageBatch], k[kImageBatch];
  const float *w[kImageBatch];
  unsigned short *img[kImageBatch], *img_t[kImageBatch];
};

__host__ __device__ inline int pad64(int v) { return (v + 63) & ~63; }

__global__ void __launch_bounds__(256) weight_images_kernel(const ImageBatch t) {
  __shared__ unsigned short tile[3][64][66];
  int e = 0;
  while (e + 1 < t.n && (int)blockIdx.x >= t.first_tile[e + 1]) ++e;
  const int local = (int)blockIdx.x - t.first_tile[e];
  const int m = t.m[e], k = t.k[e], mp = pad64(m), kp = pad64(k);
  const int tiles_k = kp / 64;
  const int r0 = (local / tiles_k) * 64, c0 = (local % tiles_k) * 64;
  const float *w = t.w[e];
  const size_t plane = (size_t)mp * kp;
  const int tid = threadIdx.x;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int rr = i >> 6, cc = i & 63;
    const float x = (r0 + rr < m && c0 + cc < k) ? w[(size_t)(r0 + rr) * k + c0 + cc] : 0.f;
    const float h = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
    const float r1 = x - h;                                   // exact
    const float md = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
    const float lo = r1 - md;                                 // exact, at most 8 significant bits
    tile[0][rr][cc] = (unsigned short)(__builtin_bit_cast(unsigned, h) >> 16);
    tile[1][rr][cc] = (unsigned short)(__builtin_bit_cast(unsigned, md) >> 16);
    tile[2][rr][cc] = (unsigned short)(__builtin_bit_cast(unsigned, lo) >> 16);
  }
  __syncthreads();
  for (int i = tid; i < 3 * 64 * 64; i += 256) {
    const int pl = i >> 12, rr = (i >> 6) & 63, cc = i & 63;
    t.img[e][pl * plane + (size_t)(r0 + rr) * kp + c0 + cc] = tile[pl][rr][cc];     // [m][k]
    t.img_t[e][pl * plane + (size_t)(c0 + rr) * mp + r0 + cc] = tile[pl][cc][rr];   // [k][m]
  }
}

}  // namespace

#define MLP_API extern "C" __attribute__((visibility("default")))

// elements (2 bytes each) of ONE image of an (m, k) weight: three planes, both dimensions padded to 64
MLP_API size_t mlp_weight_image_elems(int m, int k) {
  if (m <= 0 || k <= 0) return 0;
  return (size_t)3 * pad64(m) * pad64(k);
}

// n weights w[i] (m[i], k[i]) -> img[i] (forward order) and img_t[i] (transposed), each
// mlp_weight_image_elems(m[i], k[i]) 2-byte elements, 16-byte aligned.  The arrays are HOST arrays
// of DEVICE pointers / sizes; one launch per 24 weights.
MLP_API int mlp_weight_images_build(int n, const void *const *w, const int *m, const int *k,
                                    void *const *img, void *const *img_t, void *stream_) {
  if (n <= 0) return 0;
  if (!w || !m || !k || !img || !img_t) return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  for (int base = 0; base < n; base += kImageBatch) {
    ImageBatch t = {};
    t.n = n - base < kImageBatch ? n - base : kImageBatch;
    int tiles = 0;
    for (int i = 0; i < t.n; ++i) {
      const int j = base + i;
      if (m[j] <= 0 || k[j] <= 0 || !w[j] || !img[j] || !img_t[j] ||
          (reinterpret_cast<size_t>(img[j]) & 15) || (reinterpret_cast<size_t>(img_t[j]) & 15))
        return (int)hipErrorInvalidValue;
      t.first_tile[i] = tiles;
      t.m[i] = m[j]; t.k[i] = k[j];
      t.w[i] = reinterpret_cast<const float *>(w[j]);
      t.img[i] = reinterpret_cast<unsigned short *>(img[j]);
      t.img_t[i] = reinterpret_cast<unsigned short *>(img_t[j]);
      tiles += (pad64(m[j]) / 64) * (pad64(k[j]) / 64);
    }
    t.first_tile[t.n] = tiles;
    hipLaunchKernelGGL(weight_images_kernel, dim3(tiles), dim3(256), 0, stream, t);
  }
  return pn2_launch_status();
}

// mode: 0 = X given directly, 1 = X = relu(bn(Yprev)) via (scale, shift)
int mlp_reduce_partials(int count, int parts, const float *part, float *out, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(pn2_ceil_div((long long)count, 32)), dim3(256), 0,
                     stream, count, parts, part, out);
  return pn2_launch_status();
}

// ---- queued weight-gradient reductions (mlp_operand.h) ------------------------------------------
// One process-wide queue: the backward pass runs on autograd's worker thread, the flush on the
// caller's.  Entries of one flush share a stream (a launch on another stream flushes first).
namespace {
std::mutex reduce_queue_mutex;
bool reduce_deferred = false;
ReduceBatch reduce_queue = {};
hipStream_t reduce_queue_stream = nullptr;

int reduce_queue_launch() {  // caller holds the mutex
  if (reduce_queue.n == 0) return 0;
  const int blocks = reduce_queue.first_block[reduce_queue.n];
  hipLaunchKernelGGL(reduce_partials_batch_kernel, dim3(blocks), dim3(256), 0, reduce_queue_stream,
                     reduce_queue);
  reduce_queue.n = 0;
  return pn2_launch_status();
}
}  // namespace

int mlp_reduce_weight_partials(int count, int parts, const float *part, float *out, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(reduce_queue_mutex);
  if (!reduce_deferred) return mlp_reduce_partials(count, parts, part, out, stream);
  if (reduce_queue.n > 0 && (reduce_queue_stream != stream || reduce_queue.n == kReduceBatch)) {
    const int rc = reduce_queue_launch();
    if (rc) return rc;
  }
  const int e = reduce_queue.n++;
  if (e == 0) reduce_queue.first_block[0] = 0;
  reduce_queue_stream = stream;
  reduce_queue.count[e] = count;
  reduce_queue.parts[e] = parts;
  reduce_queue.part[e] = part;
  reduce_queue.out[e] = out;
  const bool vec = count % 4 == 0 && ((reinterpret_cast<size_t>(part) | reinterpret_cast<size_t>(out)) & 15) == 0;
  reduce_queue.vec[e] = vec ? 1 : 0;
  reduce_queue.first_block[e + 1] = reduce_queue.first_block[e] + (int)pn2_ceil_div((long long)count, vec ? 128 : 32);
  return 0;
}

// on != 0: queue the weight-gradient reductions issued from now on; on == 0: back to immediate
// launches (anything still queued is launched first).  Returns 0 or a HIP error.
MLP_API int mlp_defer_weight_reductions(int on) {
  std::lock_guard<std::mutex> lock(reduce_queue_mutex);
  int rc = 0;
  if (!on) rc = reduce_queue_launch();
  reduce_deferred = on != 0;
  return rc;
}

// launch what is queued (one kernel, on the stream the entries were issued on)
MLP_API int mlp_flush_weight_reductions(void) {
  std::lock_guard<std::mutex> lock(reduce_queue_mutex);
  return reduce_queue_launch();
}

MLP_API int mlp_gemm_forward(int b, int m, int k, int r, const float *w, const float *x, int mode,
                             const float *scale, const float *shift, float *y, void *stream_) {
  if (b <= 0 || m <= 0 || k <= 0 || r <= 0) return 0;
  OperandB op = {x, nullptr, scale, shift, nullptr, nullptr, nullptr};
  const size_t in_stride = (size_t)k * r, out_stride = (size_t)m * r;
  if (mode == OP_DIRECT)
    return launch_nn<OP_DIRECT>(b, m, k, r, w, k, op, y, in_stride, out_stride, (hipStream_t)stream_);
  return launch_nn<OP_BNRELU>(b, m, k, r, w, k, op, y, in_stride, out_stride, (hipStream_t)stream_);
}

// 1 when the layer runs on the small 64 x 64 kernel in its bf16 form, i.e. when
// mlp_gemm_forward_img / mlp_gemm_backward_small_img can take their weight from an image
MLP_API int mlp_gemm_image_supported(int b, int r) {
  static const bool off = getenv("MLP_WEIGHT_IMAGES") && atoi(getenv("MLP_WEIGHT_IMAGES")) == 0;
  const char *env = getenv("MLP_SMALL_GEMM_COLS");
  const long long small_cols = env ? atoll(env) : 16384;
  return !off && b > 0 && r > 0 && (long long)b * r <= small_cols && gemm_x6();
}

// mlp_gemm_forward with the weight ALSO given as the bf16 image mlp_weight_images_build made of it
// (img: the [plane][m][k] order): same result bit for bit, the kernel neither stages nor splits w
MLP_API int mlp_gemm_forward_img(int b, int m, int k, int r, const float *w, const void *img,
                                 const float *x, int mode, const float *scale, const float *shift,
                                 float *y, void *stream_) {
  if (b <= 0 || m <= 0 || k <= 0 || r <= 0) return 0;
  if (!img || !mlp_gemm_image_supported(b, r)) return (int)hipErrorInvalidValue;
  OperandB op = {x, nullptr, scale, shift, nullptr, nullptr, nullptr};
  const size_t in_stride = (size_t)k * r, out_stride = (size_t)m * r;
  const AImage ai = {reinterpret_cast<const unsigned short *>(img), pad64(k), (size_t)pad64(m) * pad64(k)};
  if (mode == OP_DIRECT)
    return launch_nn<OP_DIRECT>(b, m, k, r, w, k, op, y, in_stride, out_stride, (hipStream_t)stream_,
                                nullptr, ai);
  return launch_nn<OP_BNRELU>(b, m, k, r, w, k, op, y, in_stride, out_stride, (hipStream_t)stream_,
                              nullptr, ai);
}

// Can the forward GEMM of this shape leave BatchNorm partials behind?  Returns the number of
// (mean, M2) pairs per channel (0: no -- use mlp_bn_train_stats on y) and the columns each covers.
MLP_API int mlp_gemm_forward_stats_parts(int b, int m, int k, int r, int *cols_per_part) {
  const char *v2env = getenv("MLP_GEMM_PIPELINED");
  const char *stenv = getenv("MLP_GEMM_EPILOGUE_STATS");
  if ((v2env && atoi(v2env) == 0) || (stenv && atoi(stenv) == 0)) return 0;
  const char *env = getenv("MLP_SMALL_GEMM_COLS");
  const long long small_cols = env ? atoll(env) : 16384;
  if (b <= 0 || r % 256 != 0 || (long long)b * r <= small_cols) return 0;
  int tn;
  if (gemm_x6() && mlp_fwd128_enabled_for(b, m, k, r)) tn = 64;  // (128, 128): the persistent T-form kernel
  else if (m == 256) tn = 64;            // one 256 x 64 tile per column block
  else if (m > 32 && m <= 128) tn = 128; // one 128 x 128 / 64 x 128 tile
  else return 0;                         // several row tiles of different widths: not covered
  if (cols_per_part) *cols_per_part = tn;
  return b * (r / tn);
}

// mlp_gemm_forward that also writes the (mean, M2) pairs of every output channel per part into
// `pairs` (parts x m x 2 floats, parts from mlp_gemm_forward_stats_parts)
MLP_API int mlp_gemm_forward_stats(int b, int m, int k, int r, const float *w, const float *x,
                                   int mode, const float *scale, const float *shift, float *y,
                                   float *pairs, void *stream_) {
  if (b <= 0 || m <= 0 || k <= 0 || r <= 0) return 0;
  if (!pairs || mlp_gemm_forward_stats_parts(b, m, k, r, nullptr) == 0) return (int)hipErrorInvalidValue;
  if (gemm_x6() && mlp_fwd128_enabled_for(b, m, k, r))  // (its pairs cover 64 columns: no other kernel may run)
    return mlp_fwd128_launch(b, r, 0, mode == OP_DIRECT ? 1 : 0, w, x, scale, shift, nullptr, y, pairs, nullptr,
                             (hipStream_t)stream_);
  OperandB op = {x, nullptr, scale, shift, nullptr, nullptr, nullptr};
  const size_t in_stride = (size_t)k * r, out_stride = (size_t)m * r;
  if (mode == OP_DIRECT)
    return launch_nn<OP_DIRECT>(b, m, k, r, w, k, op, y, in_stride, out_stride,
                                (hipStream_t)stream_, pairs);
  return launch_nn<OP_BNRELU>(b, m, k, r, w, k, op, y, in_stride, out_stride, (hipStream_t)stream_,
                              pairs);
}

// 1 when mlp_gemm_forward_stats_pool covers the pooled last layer: statistics from the epilogue
// available, one row tile (m = 128 or 256), nsample 32 or 64, 16-byte aligned weight rows
MLP_API int mlp_gemm_forward_stats_pool_supported(int b, int m, int k, int r, int ns) {
  const char *env = getenv("MLP_GEMM_EPILOGUE_POOL");
  if (env && atoi(env) == 0) return 0;
  if (mlp_gemm_forward_stats_parts(b, m, k, r, nullptr) == 0) return 0;
  return (m == 128 || m == 256) && (ns == 16 || ns == 32 || ns == 64) && r % ns == 0 && k % 4 == 0;
}

// mlp_gemm_forward_stats (mode 1: x = raw output of the previous layer) that also leaves, per
// channel and group of ns columns, the raw output that wins the max-pool after BatchNorm (gamma:
// the layer's BatchNorm weight, whose sign decides between largest and smallest) and its first
// index: ext = 2 planes of b*m*(r/ns) 4-byte values.  y may be NULL: the raw output is then not
// stored (the layer leaves its statistics and extrema only)
MLP_API int mlp_gemm_forward_stats_pool(int b, int m, int k, int r, const float *w, const float *x,
                                        const float *scale, const float *shift, float *y,
                                        float *pairs, int ns, const float *gamma, float *ext,
                                        void *stream_) {
  if (!mlp_gemm_forward_stats_pool_supported(b, m, k, r, ns) || !pairs || !ext || !gamma ||
      (reinterpret_cast<size_t>(w) & 15) != 0)
    return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  if (gemm_x6() && mlp_fwd128_enabled_for(b, m, k, r))  // (128, 128): 64-column pairs, no other kernel may run
    return mlp_fwd128_launch(b, r, ns, 0, w, x, scale, shift, gamma, y, pairs, ext, stream);
  // (256, 128), nsample 16 / 32: the persistent T-form kernel (same pairs, same ext; y may be NULL)
  if (gemm_x6() && mlp_pool_fwd256_supported(b, m, k, r, ns, w, x) && (reinterpret_cast<size_t>(y) & 15) == 0)
    return mlp_pool_fwd256_launch(b, r, ns, w, x, scale, shift, gamma, y, pairs, ext, stream);
  OperandB op = {x, nullptr, scale, shift, nullptr, nullptr, nullptr};
  const size_t in_stride = (size_t)k * r, out_stride = (size_t)m * r;
  const size_t plane = (size_t)b * m * (r / ns);
  const unsigned a_bytes = (unsigned)(4 * (size_t)m * k);
#define POOLED_X(TM, TN, WM, WN, NS, X6)                                                           \
  hipLaunchKernelGGL((gemm_nn2_kernel<TM, TN, WM, WN, OP_BNRELU, false, true, true, NS, X6>),      \
                     dim3(r / TN, 1, b), dim3(256), 0, stream, m, k, r, w, k, a_bytes, op, y,     \
                     in_stride, out_stride, pairs, m, ext, plane, gamma)
#define POOLED(TM, TN, WM, WN, NS)                                                                 \
  do { if (gemm_x6()) POOLED_X(TM, TN, WM, WN, NS, true); else POOLED_X(TM, TN, WM, WN, NS, false); } while (0)
  if (m == 256 && ns == 16) POOLED(256, 64, 4, 1, 16);
  else if (m == 256 && ns == 32) POOLED(256, 64, 4, 1, 32);
  else if (m == 256) POOLED(256, 64, 4, 1, 64);
  else if (ns == 16) POOLED(128, 128, 2, 2, 16);
  else if (ns == 32) POOLED(128, 128, 2, 2, 32);
  else POOLED(128, 128, 2, 2, 64);
#undef POOLED
#undef POOLED_X
  return pn2_launch_status();
}

// mlp_gemm_forward_stats for the SECOND layer of a chain whose first layer has a 4-channel
// input: the operand relu(bn(W1 x4)) is recomputed from x4 (b,4,r) -- the first layer's output
// is never stored.  w (64,64), w1 (64,4), scale / shift (64) of the first layer's BatchNorm.
MLP_API int mlp_gemm_forward_stats_lin4(int b, int r, const float *w, const float *x4,
                                        const float *w1, const float *scale, const float *shift,
                                        float *y, float *pairs, void *stream_) {
  const int m = 64, k = 64;
  if (b <= 0 || r <= 0) return 0;
  if (!pairs || mlp_gemm_forward_stats_parts(b, m, k, r, nullptr) == 0 ||
      (reinterpret_cast<size_t>(w) & 15) != 0 || (reinterpret_cast<size_t>(w1) & 15) != 0)
    return (int)hipErrorInvalidValue;
  OperandB op = {x4, nullptr, scale, shift, nullptr, nullptr, nullptr, nullptr, 0, 0, w1};
  if (gemm_x6())
    hipLaunchKernelGGL((gemm_nn2_kernel<64, 128, 2, 2, OP_LIN4, false, true, true, 0, true>),
                       dim3(r / 128, 1, b), dim3(256), 0, (hipStream_t)stream_, m, k, r, w, k,
                       (unsigned)(4 * (size_t)m * k), op, y, (size_t)4 * r, (size_t)m * r, pairs, m);
  else
    hipLaunchKernelGGL((gemm_nn2_kernel<64, 128, 2, 2, OP_LIN4, false, true, true>),
                       dim3(r / 128, 1, b), dim3(256), 0, (hipStream_t)stream_, m, k, r, w, k,
                       (unsigned)(4 * (size_t)m * k), op, y, (size_t)4 * r, (size_t)m * r, pairs, m);
  return pn2_launch_status();
}

// dX (b,k,r) = W^T (k x m, given as wt row-major) * dY, with dY either given (mode 0: dy) or
// formed on the fly from (y, dz) and the per-channel vectors of the BN+ReLU backward (mode 2)
MLP_API int mlp_gemm_dgrad(int b, int m, int k, int r, const float *wt, int mode, const float *dy,
                           const float *y, const float *dz, const float *scale,
                           const float *shift, const float *mean, const float *invstd,
                           const float *coef, float *dx, void *stream_) {
  if (b <= 0 || m <= 0 || k <= 0 || r <= 0) return 0;
  const size_t in_stride = (size_t)m * r, out_stride = (size_t)k * r;
  if (mode == OP_DIRECT) {
    OperandB op = {dy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return launch_nn<OP_DIRECT>(b, k, m, r, wt, m, op, dx, in_stride, out_stride, (hipStream_t)stream_);
  }
  OperandB op = {y, dz, scale, shift, mean, invstd, coef};
  return launch_nn<OP_DY>(b, k, m, r, wt, m, op, dx, in_stride, out_stride, (hipStream_t)stream_);
}

// mlp_gemm_dgrad with the weight as stored, w (m,k) row-major: no transposed copy is needed
MLP_API int mlp_gemm_dgrad_nt(int b, int m, int k, int r, const float *w, int mode,
                              const float *dy, const float *y, const float *dz,
                              const float *scale, const float *shift, const float *mean,
                              const float *invstd, const float *coef, float *dx, void *stream_) {
  if (b <= 0 || m <= 0 || k <= 0 || r <= 0) return 0;
  const size_t in_stride = (size_t)m * r, out_stride = (size_t)k * r;
  if (mode == OP_DIRECT) {
    OperandB op = {dy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return launch_nn<OP_DIRECT, true>(b, k, m, r, w, k, op, dx, in_stride, out_stride,
                                      (hipStream_t)stream_);
  }
  OperandB op = {y, dz, scale, shift, mean, invstd, coef};
  return launch_nn<OP_DY, true>(b, k, m, r, w, k, op, dx, in_stride, out_stride,
                                (hipStream_t)stream_);
}

static bool wgrad_direct_ok(int b, int m, int k, int r, int pmode, int qmode);
static int wgrad_direct_r_per_slice(int b, int m, int k, int r);

// Both backward GEMMs of a small layer in ONE launch (gemm_small_backward_pair_kernel): 1 when
// the layer is in the small regime of both kernels.
MLP_API int mlp_gemm_backward_small_supported(int b, int m, int k, int r, int pmode, int qmode) {
  static const bool off = getenv("MLP_SMALL_BWD_PAIR") && atoi(getenv("MLP_SMALL_BWD_PAIR")) == 0;
  const char *env = getenv("MLP_SMALL_GEMM_COLS");
  const long long small_cols = env ? atoll(env) : 16384;
  return !off && b > 0 && m > 0 && k > 0 && (long long)b * r <= small_cols &&
         wgrad_direct_ok(b, m, k, r, pmode, qmode);
}

// dq (b,k,r) = w^T P[b] and dw (m,k) = sum_b P[b] Q[b]^T; P = dy (pmode 0) or formed on the fly
// from (y, dz) and the BatchNorm / ReLU backward constants (pmode 2); Q = x (qmode 0) or
// relu(x*xscale + xshift) (qmode 1).  dq == NULL: the weight gradient alone.
// workspace: mlp_gemm_wgrad_workspace_floats(b, m, k, r) floats.
static int backward_small_impl(int b, int m, int k, int r, const float *w, const void *wt_img, int pmode,
                               const float *dy_or_y, const float *dz, const float *scale,
                               const float *shift, const float *mean, const float *invstd,
                               const float *coef, int qmode, const float *x,
                               const float *xscale, const float *xshift, float *dq, float *dw,
                               float *workspace, void *stream_) {
  if (!mlp_gemm_backward_small_supported(b, m, k, r, pmode, qmode)) return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  OperandB P = {dy_or_y, dz, scale, shift, mean, invstd, coef};
  if (pmode == OP_DIRECT) P = OperandB{dy_or_y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  OperandB Q = {x, nullptr, xscale, xshift, nullptr, nullptr, nullptr};
  SmallPairArgs t;
  t.m = m; t.k = k; t.r = r;
  t.dgx = pn2_ceil_div(r, 64); t.dgy = pn2_ceil_div(k, 64);
  t.nd = dq ? t.dgx * t.dgy * b : 0;
  t.per = wgrad_direct_r_per_slice(b, m, k, r);
  t.slices = (r + t.per - 1) / t.per;
  t.wkx = pn2_ceil_div(k, 64); t.wmy = pn2_ceil_div(m, 64);
  t.w = w; t.dq = dq; t.part = workspace;
  // the data-gradient half reads W^T: rows k, reduction over m
  t.wt_img = AImage{reinterpret_cast<const unsigned short *>(wt_img), pad64(m), (size_t)pad64(k) * pad64(m)};
  const int nw = t.wkx * t.wmy * b * t.slices;
  const dim3 grid((unsigned)(t.nd + nw));
#define PAIR(PM, QM, X6D)                                                                        \
  hipLaunchKernelGGL((gemm_small_backward_pair_kernel<PM, QM, X6D>), grid, dim3(256), 0, stream, t, P, Q)
  if (pmode == OP_DIRECT && wt_img != nullptr) {
    if (qmode == OP_DIRECT)
      hipLaunchKernelGGL((gemm_small_backward_pair_kernel<OP_DIRECT, OP_DIRECT, true, true>), grid, dim3(256), 0, stream, t, P, Q);
    else
      hipLaunchKernelGGL((gemm_small_backward_pair_kernel<OP_DIRECT, OP_BNRELU, true, true>), grid, dim3(256), 0, stream, t, P, Q);
  }
  else if (pmode == OP_DIRECT && qmode == OP_DIRECT) PAIR(OP_DIRECT, OP_DIRECT, true);
  else if (pmode == OP_DIRECT) PAIR(OP_DIRECT, OP_BNRELU, true);
  else if (qmode == OP_DIRECT) PAIR(OP_DY, OP_DIRECT, false);
  else PAIR(OP_DY, OP_BNRELU, false);
#undef PAIR
  const int rc = pn2_launch_status();
  if (rc) return rc;
  return mlp_reduce_weight_partials(m * k, b * t.slices, workspace, dw, stream);
}

MLP_API int mlp_gemm_backward_small(int b, int m, int k, int r, const float *w, int pmode,
                                    const float *dy_or_y, const float *dz, const float *scale,
                                    const float *shift, const float *mean, const float *invstd,
                                    const float *coef, int qmode, const float *x,
                                    const float *xscale, const float *xshift, float *dq, float *dw,
                                    float *workspace, void *stream_) {
  return backward_small_impl(b, m, k, r, w, nullptr, pmode, dy_or_y, dz, scale, shift, mean, invstd, coef,
                             qmode, x, xscale, xshift, dq, dw, workspace, stream_);
}

// mlp_gemm_backward_small with the TRANSPOSED weight also given as the bf16 image
// mlp_weight_images_build made of it (img_t: the [plane][k][m] order), read by the data-gradient
// half when the gradient operand is given (pmode 0); same results bit for bit
MLP_API int mlp_gemm_backward_small_img(int b, int m, int k, int r, const float *w, const void *img_t,
                                        int pmode, const float *dy_or_y, const float *dz,
                                        const float *scale, const float *shift, const float *mean,
                                        const float *invstd, const float *coef, int qmode,
                                        const float *x, const float *xscale, const float *xshift,
                                        float *dq, float *dw, float *workspace, void *stream_) {
  if (!img_t || !mlp_gemm_image_supported(b, r)) return (int)hipErrorInvalidValue;
  return backward_small_impl(b, m, k, r, w, img_t, pmode, dy_or_y, dz, scale, shift, mean, invstd, coef,
                             qmode, x, xscale, xshift, dq, dw, workspace, stream_);
}

MLP_API int mlp_gemm_dgrad_pooled_nt(int b, int m, int k, int groups, int ns, const float *w,
                                     const float *y, const float *dpooled, const int *argmax,
                                     const float *scale, const float *shift, const float *mean,
                                     const float *invstd, const float *coef, float *dx,
                                     void *stream_) {
  if (b <= 0 || m <= 0 || k <= 0 || groups <= 0 || ns <= 0) return 0;
  const int r = groups * ns;
  const size_t in_stride = (size_t)m * r, out_stride = (size_t)k * r;
  OperandB op = {y, dpooled, scale, shift, mean, invstd, coef, argmax, ns, groups};
  return launch_nn<OP_POOLDY, true>(b, k, m, r, w, k, op, dx, in_stride, out_stride,
                                    (hipStream_t)stream_);
}

// the same for the pooled last layer of an SA module: dy from (y, dpooled, argmax) on the fly
MLP_API int mlp_gemm_dgrad_pooled(int b, int m, int k, int groups, int ns, const float *wt,
                                  const float *y, const float *dpooled, const int *argmax,
                                  const float *scale, const float *shift, const float *mean,
                                  const float *invstd, const float *coef, float *dx,
                                  void *stream_) {
  if (b <= 0 || m <= 0 || k <= 0 || groups <= 0 || ns <= 0) return 0;
  const int r = groups * ns;
  const size_t in_stride = (size_t)m * r, out_stride = (size_t)k * r;
  OperandB op = {y, dpooled, scale, shift, mean, invstd, coef, argmax, ns, groups};
  return launch_nn<OP_POOLDY>(b, k, m, r, wt, m, op, dx, in_stride, out_stride,
                              (hipStream_t)stream_);
}

// K is covered by column tiles of 256 / 192 / 128 / 64 (largest first), M by 64-row tiles, and
// R is cut into slices so that about 1024 workgroups are in flight (tiles x clouds x slices);
// every slice writes its own partial dW, reduced afterwards (deterministic, no atomics).
static int wgrad_next_tile(int k_left) {
  if (k_left > 192) return 256;
  if (k_left > 128) return 192;
  if (k_left > 64) return 128;
  return 64;
}

static int wgrad_r_per_slice(int b, int m, int k, int r) {
  // one launch per column tile, every launch with the same slicing: size it so that a launch
  // is exactly one resident round of workgroups (3 per CU, 2 for the 192-wide tile whose
  // six accumulators cost a wave of occupancy), minus 16 CUs of slack for the index-chain
  // kernels of the next batch that run beside it -- a partial second round is pure tail
  static const long long forced = getenv("MLP_WGRAD_WGS") ? atoll(getenv("MLP_WGRAD_WGS")) : 0;
  const long long target = forced > 0 ? forced : (wgrad_next_tile(k) == 192 ? 480 : 720);
  const long long tiles = (long long)pn2_ceil_div(m, 64) * b;
  long long slices = (target + tiles / 2) / tiles;
  if (slices < 1) slices = 1;
  long long per = (r + slices - 1) / slices;
  per = (per + RC - 1) / RC * RC;
  if (per < 64) per = 64;
  return (int)per;
}

MLP_API size_t mlp_gemm_wgrad_workspace_floats(int b, int m, int k, int r) {
  const int per = wgrad_r_per_slice(b, m, k, r);
  const int slices = (r + per - 1) / per;
  const int per_d = wgrad_direct_r_per_slice(b, m, k, r);
  const int slices_d = (r + per_d - 1) / per_d;  // (either kernel may run: size for both)
  return (size_t)b * (slices > slices_d ? slices : slices_d) * m * k;
}

// the direct-fragment kernel: small layers only (its operands must stay in L2: every 64 x 64 block
// of dW re-reads its rows), operand modes without the pooled gradient; MLP_WGRAD_DIRECT=0 disables
static bool wgrad_direct_ok(int b, int m, int k, int r, int pmode, int qmode) {
  static const bool off = getenv("MLP_WGRAD_DIRECT") && atoi(getenv("MLP_WGRAD_DIRECT")) == 0;
  static const long long cols = getenv("MLP_WGRAD_DIRECT_COLS") ? atoll(getenv("MLP_WGRAD_DIRECT_COLS")) : 32768;
  return !off && gemm_x6() && (pmode == OP_DIRECT || pmode == OP_DY) &&
         (qmode == OP_DIRECT || qmode == OP_BNRELU) && (long long)b * r <= cols && r >= 32;
}

static int wgrad_direct_r_per_slice(int b, int m, int k, int r) {
  // about 640 workgroups in flight (two per CU and some slack), at least 128 columns each
  const long long tiles = (long long)pn2_ceil_div(m, 64) * pn2_ceil_div(k, 64) * b;
  long long slices = (640 + tiles / 2) / tiles;
  if (slices < 1) slices = 1;
  long long per = (r + slices - 1) / slices;
  per = (per + 127) / 128 * 128;
  return (int)per;
}

static int wgrad_run(int b, int m, int k, int r, int pmode, const OperandB &P, int qmode,
                     const float *x, const float *xscale, const float *xshift, float *dw,
                     float *workspace, hipStream_t stream) {
  if (wgrad_direct_ok(b, m, k, r, pmode, qmode)) {
    const int per = wgrad_direct_r_per_slice(b, m, k, r);
    const int slices = (r + per - 1) / per;
    OperandB Q = {x, nullptr, xscale, xshift, nullptr, nullptr, nullptr};
    const dim3 grid(pn2_ceil_div(k, 64), pn2_ceil_div(m, 64), b * slices);
#define WGD(PM, QM)                                                                              \
  hipLaunchKernelGGL((gemm_wgrad_direct_kernel<PM, QM>), grid, dim3(256), 0, stream, m, k, r, per, \
                     slices, P, Q, workspace, (size_t)m * r, (size_t)k * r)
    if (pmode == OP_DIRECT && qmode == OP_DIRECT) WGD(OP_DIRECT, OP_DIRECT);
    else if (pmode == OP_DIRECT) WGD(OP_DIRECT, OP_BNRELU);
    else if (qmode == OP_DIRECT) WGD(OP_DY, OP_DIRECT);
    else WGD(OP_DY, OP_BNRELU);
#undef WGD
    const int rc = pn2_launch_status();
    if (rc) return rc;
    return mlp_reduce_weight_partials(m * k, b * slices, workspace, dw, stream);
  }
  const int per = wgrad_r_per_slice(b, m, k, r);
  const int slices = (r + per - 1) / per;
  OperandB Q = {x, nullptr, xscale, xshift, nullptr, nullptr, nullptr};
  const size_t ps = (size_t)m * r, qs = (size_t)k * r;
#define WG(PM, QM)                                                                              \
  do {                                                                                          \
    for (int kb = 0; kb < k;) {                                                                 \
      const int tk = wgrad_next_tile(k - kb);                                                   \
      const int ke = kb + tk < k ? kb + tk : k;                                                 \
      dim3 grid(1, pn2_ceil_div(m, 64), b * slices);                                            \
      if (tk == 256 && wgrad_x6())                                                              \
        hipLaunchKernelGGL((gemm_wgrad_kernel<PM, QM, 2, 4, 1, true>), grid, dim3(256), 0, stream, m, \
                           kb, ke, k, r, per, P, Q, workspace, ps, qs);                         \
      else if (tk == 256)                                                                       \
        hipLaunchKernelGGL((gemm_wgrad_kernel<PM, QM, 2, 4, 1>), grid, dim3(256), 0, stream, m, \
                           kb, ke, k, r, per, P, Q, workspace, ps, qs);                         \
      else if (tk == 192 && wgrad_x6())                                                          \
        hipLaunchKernelGGL((gemm_wgrad_kernel<PM, QM, 3, 2, 2, true>), grid, dim3(256), 0, stream, m, \
                           kb, ke, k, r, per, P, Q, workspace, ps, qs);                         \
      else if (tk == 192)                                                                       \
        hipLaunchKernelGGL((gemm_wgrad_kernel<PM, QM, 3, 2, 2>), grid, dim3(256), 0, stream, m, \
                           kb, ke, k, r, per, P, Q, workspace, ps, qs);                         \
      else if (tk == 128 && wgrad_x6())                                                          \
        hipLaunchKernelGGL((gemm_wgrad_kernel<PM, QM, 2, 2, 2, true>), grid, dim3(256), 0, stream, m, \
                           kb, ke, k, r, per, P, Q, workspace, ps, qs);                         \
      else if (tk == 128)                                                                       \
        hipLaunchKernelGGL((gemm_wgrad_kernel<PM, QM, 2, 2, 2>), grid, dim3(256), 0, stream, m, \
                           kb, ke, k, r, per, P, Q, workspace, ps, qs);                         \
      else                                                                                      \
        hipLaunchKernelGGL((gemm_wgrad_kernel<PM, QM, 2, 1, 4>), grid, dim3(256), 0, stream, m, \
                           kb, ke, k, r, per, P, Q, workspace, ps, qs);                         \
      kb = ke;                                                                                  \
    }                                                                                           \
  } while (0)
  if (pmode == OP_DIRECT && qmode == OP_DIRECT) WG(OP_DIRECT, OP_DIRECT);
  else if (pmode == OP_DIRECT) WG(OP_DIRECT, OP_BNRELU);
  else if (pmode == OP_DY && qmode == OP_DIRECT) WG(OP_DY, OP_DIRECT);
  else if (pmode == OP_DY) WG(OP_DY, OP_BNRELU);
  else if (qmode == OP_DIRECT) WG(OP_POOLDY, OP_DIRECT);
  else WG(OP_POOLDY, OP_BNRELU);
#undef WG
  return mlp_reduce_weight_partials(m * k, b * slices, workspace, dw, stream);
}

// dW (m x k) = sum_b dY[b] * X[b]^T; dY given (pmode 0) or on the fly (pmode 2, from y/dz);
// X given (qmode 0) or relu(bn(Yprev)) (qmode 1, via xscale/xshift).
MLP_API int mlp_gemm_wgrad(int b, int m, int k, int r, int pmode, const float *dy, const float *y,
                           const float *dz, const float *scale, const float *shift,
                           const float *mean, const float *invstd, const float *coef, int qmode,
                           const float *x, const float *xscale, const float *xshift, float *dw,
                           float *workspace, void *stream_) {
  if (b <= 0 || m <= 0 || k <= 0 || r <= 0) return 0;
  OperandB P = pmode == OP_DIRECT ? OperandB{dy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}
                                  : OperandB{y, dz, scale, shift, mean, invstd, coef};
  return wgrad_run(b, m, k, r, pmode == OP_DIRECT ? OP_DIRECT : OP_DY, P, qmode, x, xscale, xshift,
                   dw, workspace, (hipStream_t)stream_);
}

// the same for the pooled last layer: dY from (y, dpooled, argmax) on the fly
MLP_API int mlp_gemm_wgrad_pooled(int b, int m, int k, int groups, int ns, const float *y,
                                  const float *dpooled, const int *argmax, const float *scale,
                                  const float *shift, const float *mean, const float *invstd,
                                  const float *coef, int qmode, const float *x,
                                  const float *xscale, const float *xshift, float *dw,
                                  float *workspace, void *stream_) {
  if (b <= 0 || m <= 0 || k <= 0 || groups <= 0 || ns <= 0) return 0;
  OperandB P = {y, dpooled, scale, shift, mean, invstd, coef, argmax, ns, groups};
  return wgrad_run(b, m, k, groups * ns, OP_POOLDY, P, qmode, x, xscale, xshift, dw, workspace,
                   (hipStream_t)stream_);
}
