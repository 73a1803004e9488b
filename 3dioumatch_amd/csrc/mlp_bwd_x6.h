// 3dioumatch_amd/csrc/mlp_bwd_x6.h (included by mlp_bwd_fused.hip) -- the one-pass backward of a shared-MLP layer (see
// mlp_bwd_fused.hip for what it computes: dQ = W^T P and dW = sum P Q^T from ONE staging of the
// chunk) with its fp32 products on the bf16 matrix pipe (mlp_operand.h: exact three-term split,
// six of the nine partial products, v_mfma_f32_32x32x16_bf16).
//
// The fp32 kernel gathers a fragment as single floats from odd-pitch fp32 tiles; splitting those
// fragments in registers costs more vector-ALU time than the matrix pipe saves (every wave
// splits the same P fragments again).  Here every element is split ONCE, when it is staged:
//   * LDS holds, per term (hi / mid / lo), a row-major bf16 image of the chunk: P[m][n] and
//     Q[k][n], n contiguous, row pitch 80 bytes for 32 columns (80 / 16 odd: the 16-byte
//     row-wise reads of 16 consecutive rows fall on 16 distinct bank groups).  A staging lane
//     owns 16 consecutive n of one row: two 16-byte stores per term.
//   * wgrad (reduction over n): the A operand of lane (row m, eight consecutive n) and the B
//     operand (row k of Q, the same n) are ONE ds_read_b128 per term each.
//   * dgrad (reduction over m) needs, for column n, eight consecutive m -- the transpose of the
//     image.  ds_read_b64_tr_b16 does it in the LDS crossbar: the 16 lanes of a group hand in
//     the addresses of a 4 (m) x 16 (n) block as sixteen 8-byte pieces, lane i the piece
//     (m = i >> 2, n = 4 (i & 3) ..), and lane l receives column l & 15 of the block, its four m
//     (measured on the part, tools/micro: out[l][j] = piece[(l >> 2) + 4 j][l & 3]); two reads
//     per term give the lane's eight m.  No second copy of P.
//   * W^T (the A operand of dgrad) is split once per kernel and lives in registers.
// One workgroup per CU (the images are double buffered: 120 KB at M = K = 128), persistent over
// a contiguous range of 32-column chunks; staging of chunk c+1 and the loads of chunk c+2 are
// slotted between the MFMA groups of chunk c as in the fp32 kernel.
#pragma once
#include "common.h"
#include "mlp_operand.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef short bf16x4 __attribute__((ext_vector_type(4)));

#if defined(BWDX6_ABL) && BWDX6_ABL == 2   // (timing ablation 2: no MFMAs)
#define mfma_x6(ACC, A, B) do { (ACC)[0] += __builtin_bit_cast(float, (int)(A).hi[0] + (int)(B).hi[0] + (int)(A).mid[1] + (int)(B).mid[1] + (int)(A).lo[2] + (int)(B).lo[2]); } while (0)
#endif

__device__ __forceinline__ bf16x4 lds_read_tr(const char *p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4 *)(__attribute__((address_space(3))) char *)p);
}

// sum over the 32 lanes of each half-wave (DPP, no LDS); the total lands in lanes 16..31 / 48..63
__device__ __forceinline__ float x6_half_wave_sum(float v) {
#define HW_STEP(CTRL, RM) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xf, false))
  HW_STEP(0xB1, 0xf);   // quad_perm [1,0,3,2]
  HW_STEP(0x4E, 0xf);   // quad_perm [2,3,0,1]
  HW_STEP(0x141, 0xf);  // row_half_mirror
  HW_STEP(0x140, 0xf);  // row_mirror        -> every lane: the total of its row of 16
  HW_STEP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3 -> the total of the half-wave
#undef HW_STEP
  return v;
}

// MB = M / 32 (4 or 8: a wave owns MB / 4 row blocks of dW), KB = 32-column blocks of dW / rows
// blocks of Q (K padded), KBD = 32-row blocks of dQ through the matrix cores (rows xyz ..)
// STATS: also the BatchNorm-backward sums of the layer BELOW (the one whose relu(bn(.)) output is Q),
// s1 = sum g, s2 = sum g * xhat with g = dQ * [y*sc + sh > 0], xhat = (y - mu) * is, from the dQ
// blocks in the accumulators and a raw fp32 copy of the Q rows kept next to the images (the same
// arithmetic as the separate pass bn_relu_bwd_partial over (y, dQ), which it replaces):
// stats_part (k, workgroups, 2).
template <int MB, int KB, int KBD, int PMODE, int QMODE, bool STATS>
__global__ void __launch_bounds__(256, 1)
gemm_bwd_x6_kernel(int k_total, int r, int total_chunks, int chunks_per_cloud, int xyz,
                   OperandB opp, OperandB opq, const float *__restrict__ w,
                   float *__restrict__ dq, float *__restrict__ part,
                   float *__restrict__ stats_part) {
  constexpr int M = 32 * MB, KP = 32 * KB, TN = 32;
  constexpr int RP = TN * 2 + 16;              // row pitch of an image in bytes
  constexpr int PIMG = M * RP, QIMG = KP * RP;  // bytes per term
  constexpr int RAWP = (TN + 4) * 4;           // row pitch of the raw Q copy (STATS), bytes
  constexpr int QRAW = STATS ? KP * RAWP : 0;
  constexpr int BUF = 3 * (PIMG + QIMG) + QRAW;
  constexpr int RCOFF = 2 * BUF;               // float4 (sc, sh, mu, is) per Q row (STATS)
  static_assert(2 * BUF + (STATS ? KP * 16 : 0) <= 160 * 1024, "two buffers of images must fit the LDS");
  static_assert(!STATS || QMODE == OP_BNRELU, "the sums are those of a BatchNorm+ReLU layer below");
  static_assert(MB % 4 == 0 && KBD % 4 == 0, "blocks split over four waves");
  constexpr int WMB = MB / 4, WKB = KB;        // dW blocks per wave: rows wave + 4 i, all columns
  constexpr int DK = KBD / 4;                  // dQ row blocks per wave (one column block: TN = 32)
  constexpr int DG = M / 16, WG = TN / 16;     // MFMA steps of dgrad / wgrad per chunk
  // staging: a slice is one float4 (four columns) per lane; the 8 lanes of a row cover its 128
  // bytes, so that a load instruction touches whole cache lines (32 rows per slice of 256 lanes)
  constexpr int PS = M / 32, QS = KP / 32;     // slices of P / Q per chunk
  constexpr int NS = PS + QS;
  constexpr int NG = DG + WG * WKB;            // slots between MFMA groups

  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int seg_row = tid >> 3, seg_c = (tid & 7) * 4;  // row within a slice, first column
  OperandB P = opp, Q = opq;

  RowCoef pc[PS], qc[QS];
  bool q_ok[QS];
  size_t p_lane[PS], q_lane[QS];
#pragma unroll
  for (int p = 0; p < PS; ++p) {
    pc[p] = load_row_coef<PMODE>(P, seg_row + p * 32, true);
    p_lane[p] = (size_t)(seg_row + p * 32) * r + seg_c;
  }
#pragma unroll
  for (int q = 0; q < QS; ++q) {
    const int row = seg_row + q * 32;
    q_ok[q] = row < k_total;
    qc[q] = load_row_coef<QMODE>(Q, q_ok[q] ? row : k_total - 1, true);
    q_lane[q] = (size_t)(q_ok[q] ? row : k_total - 1) * r + seg_c;
  }
  // pooled form: the lane's four columns lie in group (col0 / ns) + lane_g, from sample
  // (col0 % ns) + lane_s on (ns a multiple of 4 that divides 32 or is divided by it)
  const int lane_g = PMODE == OP_POOLDY ? seg_c / P.ns : 0;
  const int lane_s = PMODE == OP_POOLDY ? seg_c % P.ns : 0;

  if (STATS) {
    float4 *rc = reinterpret_cast<float4 *>(lds + RCOFF);
    for (int t = tid; t < KP; t += 256)
      rc[t] = t < k_total ? make_float4(Q.scale[t], Q.shift[t], Q.mean[t], Q.invstd[t])
                          : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // per lane: column l31 of the 16 accumulator rows of every dQ block, over all chunks
  float st1[DK][STATS ? 16 : 1], st2[DK][STATS ? 16 : 1];
#pragma unroll
  for (int e = 0; e < DK; ++e)
#pragma unroll
    for (int q = 0; q < (STATS ? 16 : 1); ++q) { st1[e][q] = 0.f; st2[e][q] = 0.f; }

  // W^T fragments of this wave's dQ row blocks, split once: step s holds m = 16 s + 8 lhi + (0..7)
  Split3 wsp[DK][DG];
#pragma unroll
  for (int e = 0; e < DK; ++e) {
    const float *wc = w + xyz + 32 * (wave * DK + e) + l31;
#pragma unroll
    for (int s = 0; s < DG; ++s) {
      float w8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w8[j] = wc[(size_t)(16 * s + 8 * lhi + j) * k_total];
      wsp[e][s] = split3(w8);
    }
  }

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

  // raw operands of one chunk in registers: a slice of chunk c+1 is staged from them between two
  // MFMA groups of chunk c and reloaded with chunk c+2 right away.  (Two register sets with all
  // loads of chunk c+2 issued at the top of chunk c were measured: no faster -- the kernel runs at
  // the rate its 128-byte-per-row access pattern gets from HBM -- and 48 registers dearer.)
  constexpr int NSET = 1;
  float4 px[NSET][PS], pd[NSET][PS], qx[NSET][QS];
  int pwin[NSET][PS];
  float pdp[NSET][PS];
#pragma unroll
  for (int z = 0; z < NSET; ++z) {
#pragma unroll
    for (int p = 0; p < PS; ++p) {
      pwin[z][p] = -1; pdp[z][p] = 0.f;
      px[z][p] = make_float4(0.f, 0.f, 0.f, 0.f); pd[z][p] = px[z][p];
    }
#pragma unroll
    for (int q = 0; q < QS; ++q) qx[z][q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  struct ChunkAt { size_t p, q, grp; int s0; };
  auto chunk_at = [&](int c) {
    const int b = c / chunks_per_cloud;
    const int col0 = (c - b * chunks_per_cloud) * TN;
    ChunkAt at;
    at.p = (size_t)b * M * r + col0;
    at.q = (size_t)b * k_total * r + col0;
    at.grp = 0; at.s0 = 0;
    if (PMODE == OP_POOLDY) {
      const int g0 = col0 / P.ns;
      at.grp = (size_t)b * M * P.groups + g0;
      at.s0 = col0 - g0 * P.ns;
    }
    return at;
  };
  auto fetch_slice = [&](auto zt, int sl, const ChunkAt &at) {
    constexpr int z = decltype(zt)::value;
    if (sl < PS) {
      const int p = sl;
      px[z][p] = *reinterpret_cast<const float4 *>(P.x + at.p + p_lane[p]);
      if (PMODE == OP_DY) {
        pd[z][p] = *reinterpret_cast<const float4 *>(P.dz + at.p + p_lane[p]);
      } else if (PMODE == OP_POOLDY) {
        const size_t gi = at.grp + (size_t)(seg_row + p * 32) * P.groups + lane_g;
        pwin[z][p] = P.argmax[gi] - (at.s0 + lane_s);
        pdp[z][p] = P.dz[gi];
      }
    } else {
      const int q = sl - PS;
      qx[z][q] = *reinterpret_cast<const float4 *>(Q.x + at.q + q_lane[q]);
    }
  };
  // slice: four consecutive n of a row -> transformed, split, 8 bytes per term into the images
  auto store4 = [&](char *img, size_t term_bytes, int row, int c0, const float (&v)[4]) {
    float h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[e]) & 0xffff0000u);
      const float r1 = v[e] - h[e];
      m[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
      l[e] = r1 - m[e];
    }
    char *dst = img + (size_t)row * RP + c0 * 2;
    *reinterpret_cast<uint2 *>(dst) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
    *reinterpret_cast<uint2 *>(dst + term_bytes) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
    *reinterpret_cast<uint2 *>(dst + 2 * term_bytes) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
  };
  auto stage_slice = [&](auto zt, int sl, int buf) {
    constexpr int z = decltype(zt)::value;
    char *base = lds + (size_t)buf * BUF;
    if (sl < PS) {
      const int p = sl;
      const float xv[4] = {px[z][p].x, px[z][p].y, px[z][p].z, px[z][p].w};
      const float dv[4] = {pd[z][p].x, pd[z][p].y, pd[z][p].z, pd[z][p].w};
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dz = PMODE == OP_POOLDY ? (e == pwin[z][p] ? pdp[z][p] : 0.f) : dv[e];
        v[e] = transform<PMODE>(xv[e], dz, pc[p]);
      }
      store4(base, PIMG, seg_row + p * 32, seg_c, v);
    } else {
      const int q = sl - PS;
      const float xv[4] = {qx[z][q].x, qx[z][q].y, qx[z][q].z, qx[z][q].w};
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = q_ok[q] ? transform<QMODE>(xv[e], 0.f, qc[q]) : 0.f;
      store4(base + 3 * PIMG, QIMG, seg_row + q * 32, seg_c, v);
      if (STATS)
        *reinterpret_cast<float4 *>(base + 3 * (PIMG + QIMG) + (size_t)(seg_row + q * 32) * RAWP + seg_c * 4) =
            qx[z][q];
    }
  };

  using Z0 = std::integral_constant<int, 0>;
  using Z1 = std::integral_constant<int, 1>;
  auto clampc = [&](int c) { return c < c_hi ? c : c_hi - 1; };  // (surplus loads are never consumed)
  if (c_lo < c_hi) {
    const ChunkAt first = chunk_at(c_lo), second = chunk_at(clampc(c_lo + 1));
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) fetch_slice(Z0{}, sl, first);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      stage_slice(Z0{}, sl, 0);
      fetch_slice(Z0{}, sl, second);
    }
  }
  __syncthreads();

  // lane constants of the fragment reads
  // dgrad B (transposing): piece of lane i in its group of 16: m = 4 r + (i >> 2), n = 16 gn + 4 (i & 3)
  const int tr_off = (8 * lhi + ((lane & 15) >> 2)) * RP + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  // wgrad A / B (row-wise): row = block * 32 + l31, n = 16 s + 8 lhi
  const int rw_off = l31 * RP + 8 * lhi * 2;

  // one chunk: MFMAs on buffer `cur`; zt = the parity of chunk c (register set of chunks c, c+2)
  auto chunk = [&](auto zt, int c, int cur) {
    (void)zt;
    const char *Pc = lds + (size_t)cur * BUF, *Qc = Pc + 3 * PIMG;
    const ChunkAt ahead = chunk_at(clampc(c + 2));
    auto between = [&](int g) {
#if !defined(BWDX6_ABL) || BWDX6_ABL != 1   // (timing ablation 1: no staging / loads in the loop)
#pragma unroll
      for (int sl = g * NS / NG; sl < (g + 1) * NS / NG; ++sl) {
        stage_slice(Z0{}, sl, cur ^ 1);
        fetch_slice(Z0{}, sl, ahead);
      }
#endif
    };
    const int b = c / chunks_per_cloud;
    const int col0 = (c - b * chunks_per_cloud) * TN;

    // ---- dgrad: dQ block (rows k) = W^T (registers) * P (transposing reads), fragments two
    // steps ahead of the MFMAs
    f32x16 accD[DK];
#pragma unroll
    for (int e = 0; e < DK; ++e)
#pragma unroll
      for (int q = 0; q < 16; ++q) accD[e][q] = 0.f;
    bf16x4 pf[2][3][2];  // [ring][term][half of the eight m]
    auto frag = [&](int s, bf16x4 (&dst)[3][2]) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const char *p0 = Pc + (size_t)t * PIMG + (size_t)(16 * s) * RP + tr_off;
        dst[t][0] = lds_read_tr(p0);
        dst[t][1] = lds_read_tr(p0 + 4 * RP);
      }
    };
    frag(0, pf[0]);
    // wgrad fragments: P rows of this wave (per step), Q column blocks two ahead
    Split3 sp[WMB], sq[2];
    auto pfrag = [&](int s) {
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
        const char *p0 = Pc + (size_t)((wave + 4 * i) * 32) * RP + rw_off + 16 * s * 2;
        sp[i].hi = *reinterpret_cast<const bf16x8 *>(p0);
        sp[i].mid = *reinterpret_cast<const bf16x8 *>(p0 + PIMG);
        sp[i].lo = *reinterpret_cast<const bf16x8 *>(p0 + 2 * PIMG);
      }
    };
    auto qfrag = [&](int u, Split3 &dst) {  // u = s * WKB + j
      const int s = u / WKB, j = u % WKB;
      const char *q0 = Qc + (size_t)(j * 32) * RP + rw_off + 16 * s * 2;
      dst.hi = *reinterpret_cast<const bf16x8 *>(q0);
      dst.mid = *reinterpret_cast<const bf16x8 *>(q0 + QIMG);
      dst.lo = *reinterpret_cast<const bf16x8 *>(q0 + 2 * QIMG);
    };
#if defined(BWDX6_ABL) && BWDX6_ABL == 3   // (timing ablation 3: no dgrad)
    pfrag(0); qfrag(0, sq[0]);
#pragma unroll
    for (int s = 0; s < DG; ++s) between(s);
#else
#pragma unroll
    for (int s = 0; s < DG; ++s) {
      if (s + 1 < DG) frag(s + 1, pf[(s + 1) & 1]);
      if (s == DG - 1) {  // the first wgrad fragments, under the last dgrad MFMAs
        pfrag(0);
        qfrag(0, sq[0]);
      }
      Split3 sb;
      sb.hi = __builtin_shufflevector(pf[s & 1][0][0], pf[s & 1][0][1], 0, 1, 2, 3, 4, 5, 6, 7);
      sb.mid = __builtin_shufflevector(pf[s & 1][1][0], pf[s & 1][1][1], 0, 1, 2, 3, 4, 5, 6, 7);
      sb.lo = __builtin_shufflevector(pf[s & 1][2][0], pf[s & 1][2][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int e = 0; e < DK; ++e) mfma_x6(accD[e], wsp[e][s], sb);
      between(s);
    }
#endif
    // ---- wgrad: dW blocks += P * Q^T (row-wise reads of both images); the dQ blocks leave
    // between its first MFMA groups
#pragma unroll
    for (int u = 0; u < WG * WKB; ++u) {
      const int s = u / WKB, j = u % WKB;
#if !defined(BWDX6_ABL) || BWDX6_ABL != 4   // (timing ablation 4: no wgrad)
      if (u + 1 < WG * WKB) qfrag(u + 1, sq[(u + 1) & 1]);
#pragma unroll
      for (int i = 0; i < WMB; ++i) mfma_x6(accW[i][j], sp[i], sq[u & 1]);
      if (j == WKB - 1 && s + 1 < WG) pfrag(s + 1);  // (the MFMAs above hold their operands already)
#endif
#if defined(BWDX6_ABL) && (BWDX6_ABL == 5 || BWDX6_ABL == 3)   // (timing ablation 5: no dQ stores)
      if (false) {
#else
      if (u == 0) {
#endif
#pragma unroll
        for (int e = 0; e < DK; ++e) {
          const int kbd = wave * DK + e;
          float *dst = dq + ((size_t)b * k_total + xyz + 32 * kbd + 4 * lhi) * r + col0 + l31;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
#if defined(BWDX6_PLAIN_STORES)
            dst[(size_t)((q & 3) + 8 * (q >> 2)) * r] = accD[e][q];
#else
            __builtin_nontemporal_store(accD[e][q], &dst[(size_t)((q & 3) + 8 * (q >> 2)) * r]);
#endif
          }
        }
      }
      if (STATS && u == 0) {
        // every accumulator row of the dQ blocks: this lane holds column l31 and rows
        // 32*kbd + 4*lhi + (q&3) + 8*(q>>2)
        const float4 *rc = reinterpret_cast<const float4 *>(lds + RCOFF);
        const char *raw = Pc + 3 * (PIMG + QIMG);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ro = (q & 3) + 8 * (q >> 2);
#pragma unroll
          for (int e = 0; e < DK; ++e) {
            const int row = 32 * (wave * DK + e) + 4 * lhi + ro;
            const float4 c4 = rc[row];
            const float yv = *reinterpret_cast<const float *>(raw + (size_t)row * RAWP + l31 * 4);
            const float g = __fmaf_rn(yv, c4.x, c4.y) > 0.f ? accD[e][q] : 0.f;
            st1[e][q] += g;
            st2[e][q] = __fmaf_rn(g, (yv - c4.z) * c4.w, st2[e][q]);
          }
        }
      }
      between(DG + u);
    }
    __syncthreads();  // chunk c read by everyone, chunk c+1 staged by everyone
  };
  for (int c = c_lo; c < c_hi; c += 2) {
    chunk(Z0{}, c, 0);
    if (c + 1 < c_hi) chunk(Z1{}, c + 1, 1);
  }

  if (STATS && stats_part != nullptr) {
    // the lanes' column sums -> row sums through LDS (the image buffers are free: the K loop's last
    // barrier is behind every wave, and each wave parks and reads only its own rows)
    const int parts = (int)gridDim.x;
    float2 *park = reinterpret_cast<float2 *>(lds) + (size_t)wave * (32 * DK) * 33;
#pragma unroll
    for (int e = 0; e < DK; ++e)
#pragma unroll
      for (int q = 0; q < (STATS ? 16 : 1); ++q) {
        const int lrow = 32 * e + 4 * lhi + (q & 3) + 8 * (q >> 2);
        park[lrow * 33 + l31] = make_float2(st1[e][q], st2[e][q]);
      }
    for (int t = lane; t < 32 * DK; t += kWave) {  // (same wave wrote: LDS operations complete in order)
      float a1 = 0.f, a2 = 0.f;
      for (int c2 = 0; c2 < 32; ++c2) { const float2 v = park[t * 33 + c2]; a1 += v.x; a2 += v.y; }
      const int row = 32 * wave * DK + t;
      stats_part[((size_t)row * parts + blockIdx.x) * 2] = a1;
      stats_part[((size_t)row * parts + blockIdx.x) * 2 + 1] = a2;
    }
  }
  float *out = part + (size_t)blockIdx.x * M * k_total;
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
    const int mb = wave + 4 * i;
#pragma unroll
    for (int j = 0; j < WKB; ++j) {
      const int colk = j * 32 + l31;
      if (colk >= k_total) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhi;
        out[(size_t)row * k_total + colk] = accW[i][j][q];
      }
    }
  }
}

}  // namespace

// workgroups of the bf16-split backward for a (128,128) layer of b clouds x r columns (one per CU,
// at least 8 chunks each), or 0 when that kernel does not run (switched off / not this shape)
static int mlp_bwd_x6_workgroups(int b, int m, int k, int r, int cus) {
  static const bool off = (getenv("MLP_GEMM_SPLIT_BF16") && atoi(getenv("MLP_GEMM_SPLIT_BF16")) == 0) ||
                          (getenv("MLP_BWD_SPLIT_BF16") && atoi(getenv("MLP_BWD_SPLIT_BF16")) == 0);
  if (off || !(m == 128 && k == 128) || r % 32 != 0 || b <= 0) return 0;
  const long long total = (long long)b * (r / 32);
  long long g = cus;
  if (g > total / 8) g = total / 8;
  return (int)(g < 1 ? 1 : g);
}

// Launch the bf16-split backward for the shapes it covers (xyz == 0 forms); returns -1 when the
// shape is not covered (the caller then runs the fp32 kernel), else the launch status.
// stats_part (k, workgroups, 2) or null: the BatchNorm-backward sums of the layer below.
static int mlp_bwd_x6_try(int b, int m, int k, int r, int pmode, int qmode, const OperandB &P,
                          const OperandB &Q, const float *w, float *dq, float *part,
                          float *stats_part, int cus, int *workgroups, hipStream_t stream) {
  const int g = mlp_bwd_x6_workgroups(b, m, k, r, cus);
  if (g == 0 || qmode != OP_BNRELU || dq == nullptr) return -1;
  const int total_chunks = b * (r / 32), chunks_per_cloud = r / 32;
  *workgroups = g;
  constexpr int RP = 80;
#define BWDX6(MB, KB, KBD, PM, QM, ST)                                                              \
  do {                                                                                              \
    const size_t lds_bytes = 2 * (3 * (size_t)(m + k) * RP + (ST ? (size_t)k * 144 : 0)) +         \
                             (ST ? (size_t)k * 16 : 0);                                             \
    auto kern = gemm_bwd_x6_kernel<MB, KB, KBD, PM, QM, ST>;                                        \
    static bool attr_set = false;                                                                   \
    if (!attr_set) {                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);        \
      attr_set = true;                                                                              \
    }                                                                                               \
    hipLaunchKernelGGL(kern, dim3(g), dim3(256), lds_bytes, stream, k, r, total_chunks,             \
                       chunks_per_cloud, 0, P, Q, w, dq, part, stats_part);                         \
  } while (0)
  const bool st = stats_part != nullptr;
  if (pmode == OP_DY && st) BWDX6(4, 4, 4, OP_DY, OP_BNRELU, true);
  else if (pmode == OP_DY) BWDX6(4, 4, 4, OP_DY, OP_BNRELU, false);
  else if (pmode == OP_POOLDY && st) BWDX6(4, 4, 4, OP_POOLDY, OP_BNRELU, true);
  else if (pmode == OP_POOLDY) BWDX6(4, 4, 4, OP_POOLDY, OP_BNRELU, false);
  else return -1;
#undef BWDX6
  return pn2_launch_status();
}
