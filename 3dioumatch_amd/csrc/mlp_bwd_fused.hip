// 3dioumatch_amd/csrc/mlp_bwd_fused.hip -- the backward of one shared-MLP layer in ONE pass over
// its activations (gfx950, v_mfma_f32_32x32x2_f32).
//
// What it replaces: the two autograd GEMMs of a conv(1x1) layer (pointnet2/pytorch_utils.py:70-124),
//     dgrad : dQ[b] = W^T * P[b]            (K x R)     gradient w.r.t. the layer's input
//     wgrad : dW    = sum_b P[b] * Q[b]^T   (M x K)     gradient w.r.t. the weight
// where P = the BatchNorm+ReLU backward of the incoming gradient (formed on the fly from the
// pair (y, dz), or from y and the pooled tensors of an SA module's last layer) and Q = the
// layer's input relu(bn(y_prev)) (formed on the fly from y_prev).  As two kernels
// (mlp_gemm.hip) both read the pair behind P -- 2M floats per column, the larger part of the
// traffic of either.  Here a workgroup stages a TN-column chunk of P and of Q in LDS once and
// runs BOTH contractions from it: per column 2M + K floats read and K written instead of
// 4M + K*(M/64) read and K written.  Both GEMMs sit at the HBM ridge of the fp32 matrix
// cores (~31 flop/byte), so the bytes are the time.
//
// Shape of the kernel: 256 lanes = 4 waves, persistent over a contiguous range of chunks.
//   * P chunk [TN][M+1] and Q chunk [TN][KP+1] in LDS (odd leading dimension: the same tile
//     serves as MFMA A operand, unit stride, and as B operand, strided, conflict-free).
//   * dgrad: a wave owns 32-row blocks of dQ; its W^T fragments live in registers for the whole
//     kernel (M/2 values per lane and block), the P fragments come from LDS; the finished
//     32 x 32 blocks go straight from the accumulators to HBM (128-byte row segments).
//   * wgrad: a wave owns (M/32)/4 row blocks x all K column blocks of dW, accumulated in
//     registers across all chunks of the workgroup; one partial dW per workgroup at the end,
//     reduced by mlp_reduce_partials (deterministic, no atomics).
//   * software pipeline inside the workgroup, two LDS buffers: between the MFMA groups of chunk
//     c every lane transforms its share of chunk c+1 into the other buffer and reloads its
//     registers with chunk c+2 -- staging and loading hide in the shadow of the matrix pipe,
//     one barrier per chunk.
//   * grouped inputs carry three coordinate channels in front of the features (K = 3 + 32*j):
//     the feature rows of dQ go through the matrix cores at row offset 3, the three coordinate
//     rows are 3*TN dot products per chunk on the vector ALU.
#include "common.h"
#include "mlp_operand.h"
#include "mlp_bwd_x6.h"
#include <stdlib.h>

#ifndef MLP_LIN4_OCC
#define MLP_LIN4_OCC 1  // workgroups per CU of the gated (64,64) form: at 2 (256 registers) it spills inside the chunk loop and
                        // gains nothing over the form that writes dQ; at 1: step -0.05 ms (profiles/r6_step_experiments.json)
#endif

namespace {

// X6: the products run as six bf16 MFMAs on an exact three-term split of the fp32 fragments
// (mlp_operand.h: split3 / mfma_x6): a fragment is then a lane's EIGHT consecutive reduction
// indices -- eight 4-byte LDS reads from the same conflict-free tiles -- and an MFMA step covers
// 16 of them, so a chunk is M/16 dgrad steps and TN/16 wgrad steps.
template <int MB, int KB, int KBD, int NB, int PMODE, int QMODE, int OCC, bool STATS, bool DGRAD = true,
          bool X6 = false, bool GATED = true>
__global__ void __launch_bounds__(256, OCC)
gemm_bwd_fused_kernel(int k_total, int r, int total_chunks, int chunks_per_cloud, int xyz,
                      OperandB opp, OperandB opq, const float *__restrict__ w,
                      float *__restrict__ dq, float *__restrict__ part,
                      float *__restrict__ stats_part) {
  constexpr int M = 32 * MB, KP = 32 * KB, TN = 32 * NB;
  constexpr int LDP = M + 1, LDQ = KP + 1;
  constexpr int TPR = TN / 16;   // lanes per row: a lane loads 16 consecutive columns
  constexpr int RPP = 256 / TPR; // rows per pass of the 256 lanes
  constexpr int PP = (M + RPP - 1) / RPP, QP = (KP + RPP - 1) / RPP;
  // wgrad: blocks of dW per wave
  constexpr int WMB = MB >= 4 ? MB / 4 : 1;
  constexpr int WKB = MB >= 4 ? KB : KB * MB / 4;
  static_assert(MB >= 4 ? MB % 4 == 0 : (MB == 2 && KB % 2 == 0), "dW blocks must split over 4 waves");
  // dgrad: blocks of dQ per wave (DK row blocks x DN column blocks of the chunk)
  constexpr int DK = KBD >= 4 ? KBD / 4 : 1;
  constexpr int DN = KBD >= 4 ? NB : 1;
  static_assert(KBD >= 4 ? KBD % 4 == 0 : (KBD == 2 && NB == 2), "dQ blocks must split over 4 waves");
  // MFMA groups of one chunk: DG of dgrad (DU reduction steps each), WG of wgrad (WU steps each)
  constexpr int DU = 4, DG = DGRAD ? (X6 ? M / 16 : (M / 2) / DU) : 0;  // DGRAD false: the weight gradient only
  constexpr int WU = WMB * WKB >= 4 ? 1 : 2, WG = X6 ? TN / 16 : (TN / 2) / WU;
  // staging slices of one chunk per lane: a float4 (pair) of a P row, then of a Q row
  constexpr int NS = 4 * (PP + QP), NG = DG + WG;

  __shared__ float Ps[2][TN * LDP];
  __shared__ float Qs[2][TN * LDQ];
  // BatchNorm-backward sums of the layer BELOW (the one that produced Q) from the dQ blocks:
  // s1 = sum g, s2 = sum g * xhat, g = dQ * [y*sc + sh > 0], xhat = (y - mu) * is
  static_assert(DGRAD || !STATS, "the sums come from the dQ blocks");
  static_assert(!STATS || QMODE == OP_BNRELU || QMODE == OP_LIN4,
                "the sums are those of a BatchNorm+ReLU layer below");
  // QLIN: the layer below is a 4 -> K first layer whose output is never stored; its rows are
  // recomputed from the 4-channel input x4 (b,4,r) (raw: STATS is on, the tile holds raw rows)
  constexpr bool QLIN = QMODE == OP_LIN4;
  static_assert(!QLIN || (STATS && KB * 32 == 64), "recomputed rows: the 64-channel layer only");
  // QLIN: the layer below is VIRTUAL, so nobody reads dQ but (i) its BatchNorm-backward sums and (ii)
  // its weight gradient, dW1[k][c] ~ G[k][c] = sum_n gate[k][n] dQ[k][n] x4[c][n] (mlp_first4.hip).
  // Both are taken from the blocks in the accumulators and dQ (268 MB at SA1) is NOT WRITTEN: the
  // dgrad runs in "T form" -- the MFMA's two operands swapped, same fragments -- so that a lane owns
  // ONE row k of the block and 16 of its columns: the sums and the four G entries of the row are
  // in-lane accumulations (2 + 4 registers instead of 32), the column's x4 a broadcast LDS read.
  // `dq` then receives the G partials: (2 gridDim.x, 64, 4) floats.
  // (GATED = false: the round-5 form -- dQ written, the sums only -- kept for A/B, MLP_LIN4_GATED=0)
  constexpr bool TF = QLIN && GATED;
  constexpr bool RAWQ = STATS;  // the Q tile then holds raw rows, rectified as fragments are read
  constexpr int SROWS = STATS ? 32 * KBD : 1;
  __shared__ float4 Rc[SROWS];       // per Q row: sc, sh, mu, is
  __shared__ float4 X4s[TF ? 2 : 1][TF ? TN : 1];  // TF: the four input channels of every column of a chunk
  __shared__ float Wx[3 * M];        // the coordinate columns of W (xyz == 3)
  __shared__ float red[8 * 3 * 32];  // partial dot products of the coordinate rows

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int seg_row = tid / TPR, seg_c = (tid % TPR) * 16;
  OperandB P = opp, Q = opq;

  // per-row constants of the rows this lane loads (fixed for the whole kernel).  Every P row of a
  // pass exists (M is a multiple of the rows per pass); Q rows past k_total (the padding of the
  // 3 + 32j shapes) are loaded from the last real row and staged as zeros, so that the loop
  // below has no divergent branch around a load (the wait counters stay exact).
  static_assert(M % RPP == 0, "whole passes over the P rows");
  RowCoef pc[PP], qc[QP];
  bool q_ok[QP];
  size_t p_lane[PP], q_lane[QP];  // element offset of the lane's 16 columns within a cloud
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    pc[p] = load_row_coef<PMODE>(P, seg_row + p * RPP, true);
    p_lane[p] = (size_t)(seg_row + p * RPP) * r + seg_c;
  }
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int row = seg_row + q * RPP;
    q_ok[q] = row < k_total;
    qc[q] = load_row_coef<QLIN ? OP_BNRELU : QMODE>(Q, q_ok[q] ? row : k_total - 1, true);
    q_lane[q] = QLIN ? (size_t)seg_c : (size_t)(q_ok[q] ? row : k_total - 1) * r + seg_c;
  }
  float4 qw[QP];  // QLIN: the first layer's weight row of each Q row this lane stages
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int row = seg_row + q * RPP;
    qw[q] = QLIN && row < k_total ? *reinterpret_cast<const float4 *>(Q.lin_w + (size_t)row * 4)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (STATS) {
    for (int t = tid; t < SROWS; t += 256)
      Rc[t] = make_float4(Q.scale[t], Q.shift[t], Q.mean[t], Q.invstd[t]);
  }
  // with STATS the Q tile holds the RAW rows of the layer below (the sums need them) and the
  // wgrad fragments are rectified as they are read: constants of the rows kb*32 + l31
  float fsc[WKB], fsh[WKB];
#pragma unroll
  for (int j = 0; j < WKB; ++j) {
    const int kb = MB >= 4 ? j : (wave / MB) + (4 / MB) * j;
    const int row = kb * 32 + l31;
    fsc[j] = RAWQ && row < k_total ? Q.scale[row] : 1.f;
    fsh[j] = RAWQ && row < k_total ? Q.shift[row] : 0.f;
  }
  // pooled form: the lane's 16 columns lie in group (col0 / ns) + lane_g, from sample
  // (col0 % ns) + lane_s on (ns and TN divide one another)
  const int lane_g = PMODE == OP_POOLDY ? seg_c / P.ns : 0;
  const int lane_s = PMODE == OP_POOLDY ? seg_c % P.ns : 0;

  // W^T fragments of this wave's dQ row blocks: A operand of step s = W[2s + lhi][k0 + l31]
  float wreg[DK][DGRAD ? M / 2 : 1];
#pragma unroll
  for (int e = 0; e < (DGRAD ? DK : 0); ++e) {
    const int kbd = KBD >= 4 ? wave * DK + e : (wave >> 1);
    const float *wc = w + xyz + 32 * kbd + l31;
#pragma unroll
    for (int s = 0; s < M / 2; ++s)  // X6: step S = s / 8 holds m = 16 S + 8 lhi + (s % 8)
      wreg[e][s] = wc[(size_t)(X6 ? 16 * (s >> 3) + 8 * lhi + (s & 7) : 2 * s + lhi) * k_total];
  }
  for (int t = tid; t < 3 * M; t += 256) Wx[t] = DGRAD && xyz ? w[(size_t)(t % M) * k_total + t / M] : 0.f;

  f32x16 accW[WMB][WKB];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int j = 0; j < WKB; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) accW[i][j][q] = 0.f;

  const int per = (total_chunks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = (int)blockIdx.x * per;
  const int c_hi = c_lo + per < total_chunks ? c_lo + per : total_chunks;

  // raw operands of one chunk, in registers
  float4 px[PP][4], pd[PP][4], qx[QP][4][QLIN ? 4 : 1];
  int pwin[PP];
  float pdp[PP];
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    pwin[p] = -1; pdp[p] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { px[p][i] = make_float4(0.f, 0.f, 0.f, 0.f); pd[p][i] = px[p][i]; }
  }
#pragma unroll
  for (int q = 0; q < QP; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int cc = 0; cc < (QLIN ? 4 : 1); ++cc) qx[q][i][cc] = make_float4(0.f, 0.f, 0.f, 0.f);

  // where chunk c lives (uniform values)
  struct ChunkAt { size_t p, q, grp; int s0; };
  auto chunk_at = [&](int c) {
    const int b = c / chunks_per_cloud;
    const int col0 = (c - b * chunks_per_cloud) * TN;
    ChunkAt at;
    at.p = (size_t)b * M * r + col0;
    at.q = (size_t)b * (QLIN ? 4 : k_total) * r + col0;
    at.grp = 0; at.s0 = 0;
    if (PMODE == OP_POOLDY) {
      const int g0 = col0 / P.ns;
      at.grp = (size_t)b * M * P.groups + g0;
      at.s0 = col0 - g0 * P.ns;
    }
    return at;
  };
  // slice `sl` of a chunk: global -> registers
  auto fetch_slice = [&](int sl, const ChunkAt &at) {
    if (sl < 4 * PP) {
      const int p = sl >> 2, i = sl & 3;
      px[p][i] = *reinterpret_cast<const float4 *>(P.x + at.p + p_lane[p] + 4 * i);
      if (PMODE == OP_DY) {
        pd[p][i] = *reinterpret_cast<const float4 *>(P.dz + at.p + p_lane[p] + 4 * i);
      } else if (PMODE == OP_POOLDY && i == 3) {
        const size_t gi = at.grp + (size_t)(seg_row + p * RPP) * P.groups + lane_g;
        pwin[p] = P.argmax[gi] - (at.s0 + lane_s);
        pdp[p] = P.dz[gi];
      }
    } else {
      const int q = (sl - 4 * PP) >> 2, i = (sl - 4 * PP) & 3;
      if constexpr (QLIN) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          qx[q][i][cc] = *reinterpret_cast<const float4 *>(Q.x + at.q + (size_t)cc * r + q_lane[q] + 4 * i);
      } else {
        qx[q][i][0] = *reinterpret_cast<const float4 *>(Q.x + at.q + q_lane[q] + 4 * i);
      }
    }
  };
  // slice `sl` of the chunk held in the registers: transform -> LDS buffer `buf`
  auto stage_slice = [&](int sl, int buf) {
    if (sl < 4 * PP) {
      const int p = sl >> 2, i = sl & 3;
      const int row = seg_row + p * RPP;
      const float xv[4] = {px[p][i].x, px[p][i].y, px[p][i].z, px[p][i].w};
      const float dv[4] = {pd[p][i].x, pd[p][i].y, pd[p][i].z, pd[p][i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dz = PMODE == OP_POOLDY ? (4 * i + e == pwin[p] ? pdp[p] : 0.f) : dv[e];
        Ps[buf][(seg_c + 4 * i + e) * LDP + row] = transform<PMODE>(xv[e], dz, pc[p]);
      }
    } else {
      const int q = (sl - 4 * PP) >> 2, i = (sl - 4 * PP) & 3;
      const int row = seg_row + q * RPP;
      if (q * RPP >= KP) return;                   // (static) pass beyond the padded tile
      if (KP % RPP != 0 && row >= KP) return;      // last, partial pass
      float xv[4];
      if constexpr (QLIN) {  // the first layer's raw output, recomputed
        const float c0[4] = {qx[q][i][0].x, qx[q][i][0].y, qx[q][i][0].z, qx[q][i][0].w};
        const float c1[4] = {qx[q][i][1].x, qx[q][i][1].y, qx[q][i][1].z, qx[q][i][1].w};
        const float c2[4] = {qx[q][i][2].x, qx[q][i][2].y, qx[q][i][2].z, qx[q][i][2].w};
        const float c3[4] = {qx[q][i][3].x, qx[q][i][3].y, qx[q][i][3].z, qx[q][i][3].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = lin4(qw[q], c0[e], c1[e], c2[e], c3[e]);
        if (TF && row == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) X4s[buf][seg_c + 4 * i + e] = make_float4(c0[e], c1[e], c2[e], c3[e]);
        }
      } else {
        xv[0] = qx[q][i][0].x; xv[1] = qx[q][i][0].y; xv[2] = qx[q][i][0].z; xv[3] = qx[q][i][0].w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        Qs[buf][(seg_c + 4 * i + e) * LDQ + row] =
            q_ok[q] ? (RAWQ ? xv[e] : transform<QLIN ? OP_BNRELU : QMODE>(xv[e], 0.f, qc[q])) : 0.f;
    }
  };

  constexpr int SQ = (STATS && !TF) ? 16 : 1;
  float st1[DK][SQ], st2[DK][SQ];
  float gx[TF ? DK : 1][4];  // TF: this lane's row of G
#pragma unroll
  for (int e = 0; e < (TF ? DK : 1); ++e)
#pragma unroll
    for (int k = 0; k < 4; ++k) gx[e][k] = 0.f;
  float4 trc[TF ? DK : 1];   // TF: (sc, sh, mu, is) of this lane's row
#pragma unroll
  for (int e = 0; e < (TF ? DK : 1); ++e) {
    const int row = 32 * (KBD >= 4 ? wave * DK + e : (wave >> 1)) + l31;
    trc[e] = TF ? make_float4(Q.scale[row], Q.shift[row], Q.mean[row], Q.invstd[row]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int e = 0; e < DK; ++e)
#pragma unroll
    for (int q = 0; q < SQ; ++q) { st1[e][q] = 0.f; st2[e][q] = 0.f; }

  // prologue: chunk c_lo staged in buffer 0, chunk c_lo+1 on its way into the registers
  // (chunk indices past the end are clamped: the surplus loads / stagings are never consumed)
  if (c_lo < c_hi) {
    const ChunkAt first = chunk_at(c_lo), second = chunk_at(c_lo + 1 < c_hi ? c_lo + 1 : c_lo);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) fetch_slice(sl, first);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      stage_slice(sl, 0);
      fetch_slice(sl, second);
    }
  }
  __syncthreads();

  for (int c = c_lo; c < c_hi; ++c) {
    const int cur = (c - c_lo) & 1;
    const float *Pc = Ps[cur], *Qc = Qs[cur];
    const ChunkAt ahead = chunk_at(c + 2 < c_hi ? c + 2 : c_hi - 1);
    // The matrix pipe works for 64 cycles per MFMA while the wave goes on issuing: between the
    // MFMA groups of chunk c the lane transforms its slices of chunk c+1 (registers -> the other
    // LDS buffer) and reloads the registers with chunk c+2, so staging costs no time of its own
    // (two co-resident workgroups run in lock-step and do not hide it for each other).
    auto between = [&](int g) {
#if !defined(BWDF_ABL) || BWDF_ABL != 1   // (timing ablation 1: no staging / loads inside the loop)
#pragma unroll
      for (int sl = g * NS / NG; sl < (g + 1) * NS / NG; ++sl) {
        stage_slice(sl, cur ^ 1);
        fetch_slice(sl, ahead);
      }
#endif
    };
    const int b = c / chunks_per_cloud;
    const int col0 = (c - b * chunks_per_cloud) * TN;

    // ---- dgrad: dQ block = W^T (registers) * P chunk (LDS)
    f32x16 accD[DK][DN];
    if (DGRAD) {
#pragma unroll
      for (int e = 0; e < DK; ++e)
#pragma unroll
        for (int n = 0; n < DN; ++n)
#pragma unroll
          for (int q = 0; q < 16; ++q) accD[e][n][q] = 0.f;
      if constexpr (X6) {
        // step g: m = 16 g + 8 lhi + (0..7).  The raw fragments of step g+1 are requested as soon as
        // those of step g have been split (single raw buffer: the registers are free by then)
        float bp8[DN][8];
        auto frag8 = [&](int g) {
#pragma unroll
          for (int n = 0; n < DN; ++n) {
            const int nb = KBD >= 4 ? n : (wave & 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) bp8[n][j] = Pc[(nb * 32 + l31) * LDP + 16 * g + 8 * lhi + j];
          }
        };
        frag8(0);
#pragma unroll
        for (int g = 0; g < DG; ++g) {
          Split3 sb[DN];
#pragma unroll
          for (int n = 0; n < DN; ++n) sb[n] = split3(bp8[n]);
          if (g + 1 < DG) frag8(g + 1);
#pragma unroll
          for (int e = 0; e < DK; ++e) {
            float w8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = wreg[e][g * 8 + j];
            const Split3 sa = split3(w8);
#pragma unroll
            for (int n = 0; n < DN; ++n) {
              if constexpr (TF) mfma_x6(accD[e][n], sb[n], sa);  // block[column][row]
              else mfma_x6(accD[e][n], sa, sb[n]);
            }
          }
          between(g);
        }
      } else {
      // the P fragments of group g+1 are requested before the MFMAs of group g are issued
      float bp[2][DU][DN];
      auto frag = [&](int g, float (&dst)[DU][DN]) {
#pragma unroll
        for (int u = 0; u < DU; ++u)
#pragma unroll
          for (int n = 0; n < DN; ++n) {
            const int nb = KBD >= 4 ? n : (wave & 1);
            dst[u][n] = Pc[(nb * 32 + l31) * LDP + 2 * (g * DU + u) + lhi];
          }
      };
      frag(0, bp[0]);
#pragma unroll
      for (int g = 0; g < DG; ++g) {
        if (g + 1 < DG) frag(g + 1, bp[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < DU; ++u)
#pragma unroll
          for (int e = 0; e < DK; ++e)
#pragma unroll
            for (int n = 0; n < DN; ++n)
              accD[e][n] = TF ? __builtin_amdgcn_mfma_f32_32x32x2f32(bp[g & 1][u][n], wreg[e][g * DU + u],
                                                                     accD[e][n], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[e][g * DU + u], bp[g & 1][u][n],
                                                                     accD[e][n], 0, 0, 0);
        between(g);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
#pragma unroll
      for (int e = 0; e < (TF ? 0 : DK); ++e) {  // (TF: the blocks are not written)
        const int kbd = KBD >= 4 ? wave * DK + e : (wave >> 1);
#pragma unroll
        for (int n = 0; n < DN; ++n) {
          const int nb = KBD >= 4 ? n : (wave & 1);
          float *dst = dq + ((size_t)b * k_total + xyz + 32 * kbd + 4 * lhi) * r + col0 + nb * 32 + l31;
#pragma unroll
          for (int q = 0; q < 16; ++q)
            __builtin_nontemporal_store(accD[e][n][q], &dst[(size_t)((q & 3) + 8 * (q >> 2)) * r]);
        }
      }
    }

    // The BatchNorm-backward sums of the layer below, one accumulator row (q) of every dQ block
    // at a time, slotted between the wgrad MFMA groups: this lane holds column nb*32 + l31 and
    // rows 32*kbd + 4*lhi + (q&3) + 8*(q>>2) of the block.
    auto stats_row = [&](int q) {
      const int ro = (q & 3) + 8 * (q >> 2);
      if constexpr (TF) {
        // T form: register q of a block is column 4 lhi + ro, this lane's row is 32 kbd + l31
#pragma unroll
        for (int e = 0; e < DK; ++e) {
          const int kbd = KBD >= 4 ? wave * DK + e : (wave >> 1);
#pragma unroll
          for (int n = 0; n < DN; ++n) {
            const int nb = KBD >= 4 ? n : (wave & 1);
            const int col = nb * 32 + 4 * lhi + ro;
            const float yv = Qc[col * LDQ + 32 * kbd + l31];
            const float g = __fmaf_rn(yv, trc[e].x, trc[e].y) > 0.f ? accD[e][n][q] : 0.f;
            st1[e][0] += g;
            st2[e][0] = __fmaf_rn(g, (yv - trc[e].z) * trc[e].w, st2[e][0]);
            const float4 xc = X4s[cur][col];
            gx[e][0] = __fmaf_rn(g, xc.x, gx[e][0]);
            gx[e][1] = __fmaf_rn(g, xc.y, gx[e][1]);
            gx[e][2] = __fmaf_rn(g, xc.z, gx[e][2]);
            gx[e][3] = __fmaf_rn(g, xc.w, gx[e][3]);
          }
        }
        return;
      }
#pragma unroll
      for (int e = 0; e < DK; ++e) {
        const int kbd = KBD >= 4 ? wave * DK + e : (wave >> 1);
        const float4 rc = Rc[32 * kbd + 4 * lhi + ro];
#pragma unroll
        for (int n = 0; n < DN; ++n) {
          const int nb = KBD >= 4 ? n : (wave & 1);
          const float yv = Qc[(nb * 32 + l31) * LDQ + 32 * kbd + 4 * lhi + ro];
          const float g = __fmaf_rn(yv, rc.x, rc.y) > 0.f ? accD[e][n][q] : 0.f;
          st1[e][STATS ? q : 0] += g;
          st2[e][STATS ? q : 0] = __fmaf_rn(g, (yv - rc.z) * rc.w, st2[e][STATS ? q : 0]);
        }
      }
    };
    // ---- wgrad: dW blocks += P chunk * Q chunk^T (both LDS)
    if constexpr (X6) {
      // step g: n = 16 g + 8 lhi + (0..7).  The P fragments of the wave's row blocks are split
      // first; the Q column blocks follow one at a time, the next block's raw values on their way
      // while the current block's MFMAs are issued
      float qraw[8];
      auto qfrag = [&](int g, int j) {
        const int kb = MB >= 4 ? j : (wave / MB) + (4 / MB) * j;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const float raw = Qc[(16 * g + 8 * lhi + jj) * LDQ + kb * 32 + l31];
          qraw[jj] = RAWQ ? fmaxf(__fmaf_rn(raw, fsc[j], fsh[j]), 0.f) : raw;
        }
      };
#pragma unroll
      for (int g = 0; g < WG; ++g) {
        Split3 sp[WMB];
#pragma unroll
        for (int i = 0; i < WMB; ++i) {
          const int mb = MB >= 4 ? wave + 4 * i : wave % MB;
          float praw[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) praw[jj] = Pc[(16 * g + 8 * lhi + jj) * LDP + mb * 32 + l31];
          sp[i] = split3(praw);
        }
        qfrag(g, 0);
#pragma unroll
        for (int j = 0; j < WKB; ++j) {
          const Split3 sq = split3(qraw);
          if (j + 1 < WKB) qfrag(g, j + 1);
#pragma unroll
          for (int i = 0; i < WMB; ++i) mfma_x6(accW[i][j], sp[i], sq);
        }
        between(DG + g);
        if (STATS) {
#pragma unroll
          for (int q = g * 16 / WG; q < (g + 1) * 16 / WG; ++q) stats_row(q);
        }
      }
    } else {
      float ap[2][WU][WMB], bq[2][WU][WKB];
      auto frag = [&](int g, float (&a)[WU][WMB], float (&bb)[WU][WKB]) {
#pragma unroll
        for (int u = 0; u < WU; ++u) {
          const int n = 2 * (g * WU + u) + lhi;
#pragma unroll
          for (int i = 0; i < WMB; ++i) {
            const int mb = MB >= 4 ? wave + 4 * i : wave % MB;
            a[u][i] = Pc[n * LDP + mb * 32 + l31];
          }
#pragma unroll
          for (int j = 0; j < WKB; ++j) {
            const int kb = MB >= 4 ? j : (wave / MB) + (4 / MB) * j;
            const float raw = Qc[n * LDQ + kb * 32 + l31];
            bb[u][j] = RAWQ ? fmaxf(__fmaf_rn(raw, fsc[j], fsh[j]), 0.f) : raw;
          }
        }
      };
      frag(0, ap[0], bq[0]);
#pragma unroll
      for (int g = 0; g < WG; ++g) {
        if (g + 1 < WG) frag(g + 1, ap[(g + 1) & 1], bq[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < WU; ++u)
#pragma unroll
          for (int i = 0; i < WMB; ++i)
#pragma unroll
            for (int j = 0; j < WKB; ++j)
              accW[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[g & 1][u][i], bq[g & 1][u][j],
                                                                accW[i][j], 0, 0, 0);
        between(DG + g);
        if (STATS) {
#pragma unroll
          for (int q = g * 16 / WG; q < (g + 1) * 16 / WG; ++q) stats_row(q);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- the three coordinate rows of dQ (TN == 32): dot products on the vector ALU
    if (DGRAD && NB == 1 && xyz) {
      const int n = tid & 31, g8 = tid >> 5;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 4
      for (int mm = g8 * (M / 8); mm < (g8 + 1) * (M / 8); ++mm) {
        const float pv = Pc[n * LDP + mm];
        s0 = __fmaf_rn(pv, Wx[mm], s0);
        s1 = __fmaf_rn(pv, Wx[M + mm], s1);
        s2 = __fmaf_rn(pv, Wx[2 * M + mm], s2);
      }
      red[(g8 * 3 + 0) * 32 + n] = s0;
      red[(g8 * 3 + 1) * 32 + n] = s1;
      red[(g8 * 3 + 2) * 32 + n] = s2;
      __syncthreads();
      if (tid < 96) {
        const int kx = tid >> 5;
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[(g * 3 + kx) * 32 + n];
        dq[((size_t)b * k_total + kx) * r + col0 + n] = t;
      }
    }
    __syncthreads();  // chunk c read by everyone, chunk c+1 staged by everyone
  }

  if constexpr (TF) {
    // T form: the two halves of a wave hold the same rows (other columns); one partial per row and
    // (workgroup, column block); the rows of G go where dQ would have gone
    constexpr int PPW = KBD >= 4 ? 1 : 2;
    const int parts = (int)gridDim.x * PPW;
    const int pidx = (int)blockIdx.x * PPW + (KBD >= 4 ? 0 : (wave & 1));
#pragma unroll
    for (int e = 0; e < DK; ++e) {
      const int row = 32 * (KBD >= 4 ? wave * DK + e : (wave >> 1)) + l31;
      float a1 = st1[e][0], a2 = st2[e][0];
      a1 += __shfl_xor(a1, 32, kWave);
      a2 += __shfl_xor(a2, 32, kWave);
      float g4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) g4[k] = gx[e][k] + __shfl_xor(gx[e][k], 32, kWave);
      if (lhi == 0) {
        if (stats_part != nullptr) {
          stats_part[((size_t)row * parts + pidx) * 2] = a1;
          stats_part[((size_t)row * parts + pidx) * 2 + 1] = a2;
        }
        *reinterpret_cast<float4 *>(dq + ((size_t)pidx * k_total + row) * 4) = make_float4(g4[0], g4[1], g4[2], g4[3]);
      }
    }
  } else if (STATS && stats_part != nullptr) {
    // sum over the 32 columns held by the lanes of a half-wave, one partial per row and
    // workgroup (two when two waves share a row block: KBD == 2)
    constexpr int PPW = KBD >= 4 ? 1 : 2;
    const int parts = (int)gridDim.x * PPW;
    const int pidx = (int)blockIdx.x * PPW + (KBD >= 4 ? 0 : (wave & 1));
#pragma unroll
    for (int e = 0; e < DK; ++e) {
      const int kbd = KBD >= 4 ? wave * DK + e : (wave >> 1);
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        float a1 = st1[e][q], a2 = st2[e][q];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
          a1 += __shfl_xor(a1, off, kWave);
          a2 += __shfl_xor(a2, off, kWave);
        }
        if (l31 == 0) {
          const int row = 32 * kbd + 4 * lhi + (q & 3) + 8 * (q >> 2);
          stats_part[((size_t)row * parts + pidx) * 2] = a1;
          stats_part[((size_t)row * parts + pidx) * 2 + 1] = a2;
        }
      }
    }
  }
  float *out = part + (size_t)blockIdx.x * M * k_total;
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
    const int mb = MB >= 4 ? wave + 4 * i : wave % MB;
#pragma unroll
    for (int j = 0; j < WKB; ++j) {
      const int kb = MB >= 4 ? j : (wave / MB) + (4 / MB) * j;
      const int colk = kb * 32 + l31;
      if (colk >= k_total) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhi;
        out[(size_t)row * k_total + colk] = accW[i][j][q];
      }
    }
  }
}

// one supported layer shape
struct FusedShape { int m, k, xyz, tn, occ; };

bool fused_shape(int m, int k, FusedShape *s) {
  static const FusedShape table[] = {
      {64, 64, 0, 64, 2}, {128, 64, 0, 64, 1}, {128, 128, 0, 32, 2},
      {256, 128, 0, 32, 1}, {128, 131, 3, 32, 1}, {128, 259, 3, 32, 1},
  };
  for (const FusedShape &t : table)
    if (t.m == m && t.k == k) { *s = t; return true; }
  return false;
}

int fused_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}

// workgroups of the persistent launch: one resident round, at least 8 chunks each (the partial
// dW every workgroup writes must stay small next to the activations it read)
int fused_workgroups(const FusedShape &s, long long total_chunks) {
  static const long long forced = getenv("MLP_FUSED_BWD_WGS") ? atoll(getenv("MLP_FUSED_BWD_WGS")) : 0;
  long long g = forced > 0 ? forced : (long long)fused_cus() * s.occ;
  if (g > total_chunks / 8) g = total_chunks / 8;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

#define MLP_API extern "C" __attribute__((visibility("default")))

// 1 when mlp_gemm_backward_fused covers the layer: (m,k) one of the shared-MLP shapes of the
// network, whole chunks per cloud, the pooled form with nsample a multiple of 16
MLP_API int mlp_gemm_backward_fused_supported(int b, int m, int k, int r, int pmode, int qmode,
                                              int ns) {
  static const bool off = getenv("MLP_FUSED_BACKWARD") && atoi(getenv("MLP_FUSED_BACKWARD")) == 0;
  FusedShape s;
  if (off || b <= 0 || r <= 0 || !fused_shape(m, k, &s)) return 0;
  if (r % s.tn != 0 || (long long)b * (r / s.tn) < 64) return 0;
  if (pmode != OP_DY && pmode != OP_POOLDY) return 0;
  if (pmode == OP_POOLDY && (ns <= 0 || ns % 16 != 0 || r % ns != 0 || (ns % s.tn != 0 && s.tn % ns != 0)))
    return 0;
  // instantiated operand combinations (the layers of the network)
  if (qmode == OP_LIN4) return m == 64 && k == 64 && pmode == OP_DY;  // recomputed first layer below
  const bool first = s.xyz != 0;  // grouped input: the layer reads the network input directly
  if (first != (qmode == OP_DIRECT)) return 0;
  if (qmode != OP_DIRECT && qmode != OP_BNRELU) return 0;
  // instantiated (shape, gradient operand) pairs: the pooled last layers are (128,64), (256,128)
  // and (128,128) (vote aggregation, grid features); (128,128) also occurs unpooled
  const bool pooled_only = (m == 128 && k == 64) || (m == 256 && k == 128);
  const bool either = m == 128 && k == 128;
  if (!either && pooled_only != (pmode == OP_POOLDY)) return 0;
  return 1;
}

// 1 when mlp_gemm_backward_fused with qmode 4 leaves the gated sums of the virtual layer below in dq
// (parts x 64 x 4 floats) instead of the data gradient (b, 64, r); 0 with MLP_LIN4_GATED=0 (A/B)
MLP_API int mlp_gemm_backward_fused_lin4_gated(void) {
  static const bool off = getenv("MLP_LIN4_GATED") && atoi(getenv("MLP_LIN4_GATED")) == 0;
  return off ? 0 : 1;
}

// number of (s1, s2) partials per channel that mlp_gemm_backward_fused leaves in stats_part
// (k, parts, 2) for the layer below (qmode 1 only; 0 otherwise)
MLP_API int mlp_gemm_backward_fused_stats_parts(int b, int m, int k, int r) {
  FusedShape s;
  // the (128,128) layers on the bf16-split kernel: one partial per workgroup
  const int gx = mlp_bwd_x6_workgroups(b, m, k, r, fused_cus());
  if (gx > 0) return gx;
  // otherwise the first set-abstraction level only: there the absorbed pass over (x, dq) costs
  // ~95 us a layer; on the narrower levels the extra registers cost the fp32 GEMM more than the pass
  if (!fused_shape(m, k, &s) || k != 64 || r % s.tn != 0) return 0;
  return fused_workgroups(s, (long long)b * (r / s.tn)) * 2;
}

MLP_API size_t mlp_gemm_backward_fused_workspace_floats(int b, int m, int k, int r) {
  FusedShape s;
  if (!fused_shape(m, k, &s) || r % s.tn != 0) return 0;
  // (the bf16-split kernel of the (128,128) layers writes one partial per ITS workgroup count)
  const long long g0 = fused_workgroups(s, (long long)b * (r / s.tn));
  const long long g1 = mlp_bwd_x6_workgroups(b, m, k, r, fused_cus());
  return (size_t)(g0 > g1 ? g0 : g1) * m * k;
}

// dq (b,k,r) = W^T * P[b] and dw (m,k) = sum_b P[b] * Q[b]^T in one pass.
// pmode 2: P from (y, dz) (b,m,r); pmode 3: from y, dz = dpooled (b,m,r/ns) and argmax.
// qmode 1: Q = relu(x*xscale + xshift); qmode 0: Q = x.
MLP_API int mlp_gemm_backward_fused(int b, int m, int k, int r, const float *w, int pmode,
                                    const float *y, const float *dz, const int *argmax, int ns,
                                    const float *scale, const float *shift, const float *mean,
                                    const float *invstd, const float *coef, int qmode,
                                    const float *x, const float *xscale, const float *xshift,
                                    const float *xmean, const float *xinvstd, const float *xlin_w,
                                    float *dq, float *dw, float *workspace, float *stats_part,
                                    void *stream_) {
  if (!mlp_gemm_backward_fused_supported(b, m, k, r, pmode, qmode, ns)) return (int)hipErrorInvalidValue;
  FusedShape s;
  fused_shape(m, k, &s);
  if (dq == nullptr && !(m == 128 && k == 259)) return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  const int cpc = r / s.tn;
  const int total = b * cpc;
  const int g = fused_workgroups(s, total);
  OperandB P = {y, dz, scale, shift, mean, invstd, coef, argmax, ns, ns > 0 ? r / ns : 0};
  if ((qmode == OP_BNRELU || qmode == OP_LIN4) && (!xmean || !xinvstd)) return (int)hipErrorInvalidValue;
  if (qmode == OP_LIN4 && !xlin_w) return (int)hipErrorInvalidValue;
  OperandB Q = {x, nullptr, xscale, xshift, xmean, xinvstd, nullptr, nullptr, 0, 0, xlin_w};
  {  // the bf16-split kernel with its operands split at staging, where it covers the shape
    int gx = 0;
    const int rcx = mlp_bwd_x6_try(b, m, k, r, pmode, qmode, P, Q, w, dq, workspace,
                                   qmode == OP_BNRELU ? stats_part : nullptr, fused_cus(), &gx, stream);
    if (rcx > 0) return rcx;
    if (rcx == 0) return mlp_reduce_weight_partials(m * k, gx, workspace, dw, stream);
  }
  if (k != 64) stats_part = nullptr;
  static const bool x6 = !(getenv("MLP_GEMM_SPLIT_BF16") && atoi(getenv("MLP_GEMM_SPLIT_BF16")) == 0) &&
                         !(getenv("MLP_BWD_SPLIT_BF16") && atoi(getenv("MLP_BWD_SPLIT_BF16")) == 0);
#define FUSED_X(MB, KB, KBD, NB, PM, QM, OCC, ST, X6)                                           \
  hipLaunchKernelGGL((gemm_bwd_fused_kernel<MB, KB, KBD, NB, PM, QM, OCC, ST, true, X6>), dim3(g), \
                     dim3(256), 0, stream, k, r, total, cpc, s.xyz, P, Q, w, dq, workspace,     \
                     stats_part)
#define FUSED(MB, KB, KBD, NB, PM, QM, OCC, ST)                                                 \
  do { if (x6) FUSED_X(MB, KB, KBD, NB, PM, QM, OCC, ST, true);                                 \
       else FUSED_X(MB, KB, KBD, NB, PM, QM, OCC, ST, false); } while (0)
  // (the bf16 form wins where the registers hold it; measured per shape, profiles/r4_split_bf16.json)
#define FUSED_F32(MB, KB, KBD, NB, PM, QM, OCC, ST) FUSED_X(MB, KB, KBD, NB, PM, QM, OCC, ST, false)
  if (m == 64 && k == 64 && qmode == OP_LIN4 && !mlp_gemm_backward_fused_lin4_gated()) {
    if (x6) hipLaunchKernelGGL((gemm_bwd_fused_kernel<2, 2, 2, 2, OP_DY, OP_LIN4, 2, true, true, true, false>), dim3(g),
                               dim3(256), 0, stream, k, r, total, cpc, s.xyz, P, Q, w, dq, workspace, stats_part);
    else hipLaunchKernelGGL((gemm_bwd_fused_kernel<2, 2, 2, 2, OP_DY, OP_LIN4, 2, true, true, false, false>), dim3(g),
                            dim3(256), 0, stream, k, r, total, cpc, s.xyz, P, Q, w, dq, workspace, stats_part);
  } else if (m == 64 && k == 64 && qmode == OP_LIN4) FUSED(2, 2, 2, 2, OP_DY, OP_LIN4, MLP_LIN4_OCC, true);
  else if (m == 64 && k == 64) FUSED(2, 2, 2, 2, OP_DY, OP_BNRELU, 2, true);
  else if (m == 128 && k == 64) FUSED(4, 2, 2, 2, OP_POOLDY, OP_BNRELU, 1, true);
  else if (m == 128 && k == 128 && pmode == OP_DY) FUSED_F32(4, 4, 4, 1, OP_DY, OP_BNRELU, 2, false);
  else if (m == 128 && k == 128) FUSED_F32(4, 4, 4, 1, OP_POOLDY, OP_BNRELU, 2, false);
  else if (m == 256 && k == 128) FUSED(8, 4, 4, 1, OP_POOLDY, OP_BNRELU, 1, false);
  else if (m == 128 && k == 131) FUSED(4, 5, 4, 1, OP_DY, OP_DIRECT, 1, false);
  else if (dq != nullptr) FUSED_F32(4, 9, 8, 1, OP_DY, OP_DIRECT, 1, false);
  else if (x6 && !(getenv("MLP_WGRAD_ONLY_SPLIT") && atoi(getenv("MLP_WGRAD_ONLY_SPLIT")) == 0))
    // the layer's input needs no gradient: the persistent weight-gradient half alone (its bf16 form
    // fits the registers: no W^T fragments, no dQ blocks)
    hipLaunchKernelGGL((gemm_bwd_fused_kernel<4, 9, 8, 1, OP_DY, OP_DIRECT, 1, false, false, true>), dim3(g),
                       dim3(256), 0, stream, k, r, total, cpc, s.xyz, P, Q, w, dq, workspace, stats_part);
  else
    hipLaunchKernelGGL((gemm_bwd_fused_kernel<4, 9, 8, 1, OP_DY, OP_DIRECT, 1, false, false>), dim3(g),
                       dim3(256), 0, stream, k, r, total, cpc, s.xyz, P, Q, w, dq, workspace, stats_part);
#undef FUSED_F32
#undef FUSED
#undef FUSED_X
  int rc = pn2_launch_status();
  if (rc) return rc;
  return mlp_reduce_weight_partials(m * k, g, workspace, dw, stream);
}
