// 3dioumatch_amd/csrc/mlp_pool_fwd256.hip -- forward of the max-pooled LAST shared-MLP layer of
// SA2 / SA3 / SA4 (128 -> 256 channels: conv(1x1) of pytorch_utils.py:70-124 on relu(bn(y2)), then
// the max over nsample of pointnet2_modules.py:256-262) when its raw output is NOT stored (gfx950):
// the layer leaves its BatchNorm (mean, M2) pairs and the pooled extrema only (its backward runs from
// the Gram matrix of its input, mlp_pool_gram256.hip; the teacher's pass has no backward at all).
//
// Reached through mlp_gemm_forward_stats_pool(y = NULL) (mlp_gemm.hip); same pairs (one per channel
// and 64 columns) and the same ext planes as that entry's tiled kernel.
//
// Why a kernel of its own: the tiled kernel stages fp32 tiles, and each of its four waves splits the
// SAME 64 columns of the operand into bf16 terms again (and the weight, per tile) -- it is bound by
// vector issue (156-165 us at SA2 against 45-50 us of MFMA time).  Here
//   * a workgroup is persistent: 8 waves, wave w owns channels 32 w .. 32 w + 31; its rows of W3 are
//     split ONCE into register fragments (96 registers);
//   * a chunk of y2 (128 rows x 32 columns) is transformed, split once by the thread that loaded it
//     and stored as three bf16 images (fragments by transposing LDS reads, as mlp_bwd_x6.h), in one
//     of THREE buffers: chunk c + 1 is staged while chunk c is multiplied, ONE barrier per chunk in
//     the middle of the MFMAs, and the first fragments of chunk c + 1 are requested behind chunk c's
//     last MFMAs.  Image rows are 64 bytes apart, unpadded: the transposing read of a half-wave covers
//     four consecutive rows x 64 bytes = all 64 banks once, the staging write of a wave 512 contiguous
//     bytes -- SQ_LDS_BANK_CONFLICT 0 (with the 80-byte pitch of the other kernels: half of the LDS
//     cycles);
//   * the product runs in T FORM (the MFMA's operands swapped: A = the chunk's fragment, B = W3's):
//     a lane then owns ONE channel and 16 of the chunk's 32 columns, so the BatchNorm sums and the
//     group's extremum are in-lane scans (packed fp32 arithmetic), finished by one exchange with
//     lane ^ 32 -- no transposition through LDS; what leaves: 8 bytes per channel and 64 columns, and
//     the extrema of FOUR consecutive groups as one 16-byte store per lane;
//   * everything but the MFMAs is cut into pieces of about eight vector instructions, one behind
//     every two MFMAs: the next chunk's staging, and the PREVIOUS chunk's epilogue (software
//     pipeline: its samples are copied out of the accumulators when its MFMAs have drained);
//   * two accumulators, alternating: an MFMA that accumulates into the block of the MFMA right before
//     it is forwarded, but with vector instructions between them it waits for the write-back
//     (tools/micro/mfma_bf16_peak.py: 2 waves per SIMD, 4 vector instructions per MFMA: 82 % of the
//     MFMA-only rate with one accumulator, 97 % with two).
// Measured (B = 8, SA2: 8192 chunks): 90 us against 156 (tiled kernel, store removed); MFMA-only
// variant of the same loop 66 us, MFMA peak 45.
#include "common.h"
#include "mlp_operand.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

typedef short f_bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f_bf16x4 f_lds_read_tr(const char *p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) f_bf16x4 *)(__attribute__((address_space(3))) char *)p);
}

constexpr int kFM = 256, kFK = 128, kFTN = 32;
constexpr int kFRP = kFTN * 2;        // image row pitch (bytes): 64 -- see the note on bank conflicts below
constexpr int kFIMG = kFK * kFRP;     // bytes per term
constexpr int kFBUF = 3 * kFIMG;      // one chunk: 24 576 bytes
constexpr int kFTP = 32 * 144;        // STORE form: a wave's 32 x 32 tile on its way out

struct PoolFwdArgs {
  int r, chunks_per_cloud, total_tiles, groups;
  const float *w;             // (256, 128)
  const float *x;             // (b, 128, r) raw output of the layer below
  const float *sc, *sh;       // (128) its BatchNorm as scale / shift
  const float *gamma;         // (256) this layer's BatchNorm weight: its sign picks the extremum
  float *pairs;               // (b * r / 64, 256, 2)
  float *ext;                 // 2 planes of (b, 256, groups)
  size_t ext_plane;
  float *y;                   // (b, 256, r) raw output, STORE form only
};

typedef float f_f32x2 __attribute__((ext_vector_type(2)));
typedef float f_f32x4 __attribute__((ext_vector_type(4)));

template <int NS, bool STORE>
__global__ void __launch_bounds__(512) pool_fwd256_kernel(const PoolFwdArgs a) {
  constexpr int K = kFK, TN = kFTN, RP = kFRP, IMG = kFIMG, BUF = kFBUF;
  constexpr int G = TN / NS;    // groups per chunk: 1 or 2
  constexpr int QG = 16 / G;    // accumulator registers per group
  static_assert(NS == 16 || NS == 32, "nsample");
  extern __shared__ __attribute__((aligned(16))) char lds[];  // three chunk buffers

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int seg_row = tid >> 3, seg_c = (tid & 7) * 4;  // staging: rows seg_row, seg_row + 64
  const int ch = 32 * wave + l31;

  // B fragments: this lane's channel, k = 16 s + 8 lhi + 0..7
  Split3 wsp[K / 16];
  {
    const float *wr = a.w + (size_t)ch * K + 8 * lhi;
#pragma unroll
    for (int s = 0; s < K / 16; ++s)
      wsp[s] = split3(*reinterpret_cast<const float4 *>(wr + 16 * s),
                      *reinterpret_cast<const float4 *>(wr + 16 * s + 4));
  }
  const bool neg = a.gamma[ch] < 0.f;
  const float sg = neg ? -1.f : 1.f;
  const f_f32x2 sgn = {sg, sg};
  const RowCoef rc0 = {a.sc[seg_row], a.sh[seg_row], 0.f, 0.f, 0.f};
  const RowCoef rc1 = {a.sc[seg_row + 64], a.sh[seg_row + 64], 0.f, 0.f, 0.f};

  // whole blocks of four chunks (two 64-column tiles) per workgroup: the extrema leave four groups at
  // a time
  const int blocks = a.total_tiles / 2;
  const int per = (blocks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int k_lo = (int)blockIdx.x * per;
  const int k_hi = k_lo + per < blocks ? k_lo + per : blocks;
  const int c_lo = 4 * k_lo, c_hi = 4 * k_hi;
  if (c_lo >= c_hi) return;

  // requests run TWO chunks ahead of their use (qf: the chunk after the next; it moves to qx when
  // the next chunk's staging has consumed qx): one chunk of MFMAs does not cover a miss to HBM
  float4 qx0, qx1, qf0, qf1;
  auto fetch = [&](int c) {
    const int b = c / a.chunks_per_cloud;
    const int col0 = (c - b * a.chunks_per_cloud) * TN;
    const float *src = a.x + ((size_t)b * K + seg_row) * a.r + col0 + seg_c;
    qf0 = *reinterpret_cast<const float4 *>(src);
    qf1 = *reinterpret_cast<const float4 *>(src + (size_t)64 * a.r);
  };
  auto stage_slice = [&](char *base, int row, const float4 &x, const RowCoef &rc) {
    const float xv[4] = {x.x, x.y, x.z, x.w};
    float h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = transform<OP_BNRELU>(xv[e], 0.f, rc);
      h[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u);
      const float r1 = v - h[e];
      m[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
      l[e] = r1 - m[e];
    }
    char *dst = base + (size_t)row * RP + seg_c * 2;
    *reinterpret_cast<uint2 *>(dst) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
    *reinterpret_cast<uint2 *>(dst + IMG) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
    *reinterpret_cast<uint2 *>(dst + 2 * IMG) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
  };
  auto stage = [&](char *base) {
    stage_slice(base, seg_row, qx0, rc0);
    stage_slice(base, seg_row + 64, qx1, rc1);
  };
  auto clampc = [&](int c) { return c < c_hi ? c : c_hi - 1; };

  // transposing reads: the lane's fragment = column l31 of the chunk, k = 16 s + 8 lhi + 0..7
  const int tr_off = (8 * lhi + ((lane & 15) >> 2)) * RP + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  f_bf16x4 pf[2][3][2];  // ring of two k-steps' fragments
  auto frag = [&](const char *Q, int s, f_bf16x4 (&dst)[3][2]) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const char *p0 = Q + (size_t)t * IMG + (size_t)(16 * s) * RP + tr_off;
      dst[t][0] = f_lds_read_tr(p0);
      dst[t][1] = f_lds_read_tr(p0 + 4 * RP);
    }
  };
  auto operand = [&](const f_bf16x4 (&src)[3][2]) {
    Split3 sb;
    sb.hi = __builtin_shufflevector(src[0][0], src[0][1], 0, 1, 2, 3, 4, 5, 6, 7);
    sb.mid = __builtin_shufflevector(src[1][0], src[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
    sb.lo = __builtin_shufflevector(src[2][0], src[2][1], 0, 1, 2, 3, 4, 5, 6, 7);
    return sb;
  };

  fetch(c_lo);
  qx0 = qf0; qx1 = qf1;
  stage(lds);
  fetch(clampc(c_lo + 1));
  qx0 = qf0; qx1 = qf1;
  fetch(clampc(c_lo + 2));
  __syncthreads();
  frag(lds, 0, pf[0]);

  float tsh = 0.f;
  f_f32x2 t1 = {0.f, 0.f}, t2 = {0.f, 0.f};  // shifted sums of the lane's samples of the current tile
  unsigned held[4] = {0u, 0u, 0u, 0u};       // the last four groups' values (lhi 0) / indices (lhi 1)
  int cur = 0;

  // ---- the work around the MFMAs, in PIECES of about eight vector instructions: each sits behind two
  // MFMAs of the chunk in flight with a scheduling barrier behind it (left to itself the compiler issues
  // the chunk's 48 MFMAs back to back and everything else before or after them, where the matrix pipe
  // idles: both waves of a SIMD are in step, a barrier per chunk sees to that).
  // Pieces 0-8: the NEXT chunk (c + 1: transform, split, images; the request for c + 2).
  // Pieces 9-21: the epilogue of the PREVIOUS chunk (ce = c - 1, its accumulators copied to pv).
  float sv[4], shh[4], sm[4], sl[4];  // staging state between pieces
  f_f32x2 pv[8];                      // previous chunk's samples
  float best[G], mh = 0.f, qh = 0.f, mo = 0.f, qo = 0.f;
  float ob[G];
  int at[G], oa[G];
  char *Qn = lds;
  int ce = 0;
  bool real = false;  // chunk ce exists (not the run-in of the pipeline)
  f_f32x4 yrow[2];  // STORE: two rows' pieces on their way from the LDS tile to y
  auto store_rows = [&](int k0) {
    if (real) {
      const int b = ce / a.chunks_per_cloud;
      float *dst = a.y + ((size_t)b * kFM + 32 * wave + (lane >> 3)) * a.r + (ce - b * a.chunks_per_cloud) * TN + 4 * (lane & 7);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
        __builtin_nontemporal_store(yrow[k2], reinterpret_cast<f_f32x4 *>(dst + (size_t)(8 * (k0 + k2)) * a.r));
    }
  };
  auto piece = [&](auto nt) {
    constexpr int n = decltype(nt)::value;
    if constexpr (n == 0 || n == 4) {
      const float4 &x = n == 0 ? qx0 : qx1;
      const RowCoef &rc = n == 0 ? rc0 : rc1;
      const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) sv[e] = transform<OP_BNRELU>(xv[e], 0.f, rc);
    } else if constexpr (n == 1 || n == 5) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        shh[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sv[e]) & 0xffff0000u);
        sv[e] = sv[e] - shh[e];
      }
    } else if constexpr (n == 2 || n == 6) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sm[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sv[e]) & 0xffff0000u);
        sl[e] = sv[e] - sm[e];
      }
    } else if constexpr (n == 3 || n == 7) {
      char *dst = Qn + (size_t)(seg_row + (n == 7 ? 64 : 0)) * RP + seg_c * 2;
      *reinterpret_cast<uint2 *>(dst) = make_uint2(pack_hi16(shh[0], shh[1]), pack_hi16(shh[2], shh[3]));
      *reinterpret_cast<uint2 *>(dst + IMG) = make_uint2(pack_hi16(sm[0], sm[1]), pack_hi16(sm[2], sm[3]));
      *reinterpret_cast<uint2 *>(dst + 2 * IMG) = make_uint2(pack_hi16(sl[0], sl[1]), pack_hi16(sl[2], sl[3]));
    } else if constexpr (n == 8) {
      qx0 = qf0; qx1 = qf1;
      fetch(clampc(ce + 4));
    } else if constexpr (n == 9 || n == 10 || n == 11) {
      if (n == 9 && (ce & 1) == 0) { tsh = pv[0].x; t1 = f_f32x2{0.f, 0.f}; t2 = t1; }
      if constexpr (STORE) {
        // T form hands a lane ONE channel and four runs of four consecutive columns: stored as such
        // they are 16-byte requests, one per lane (541 us at SA2 against 178).  So the block goes
        // through a wave-private LDS tile (rows 144 bytes apart) and leaves as rows: a store covers
        // eight channels x 128 contiguous bytes.  (LDS operations of one wave complete in order: no
        // barrier between its writes and reads, nor between this chunk's reads and the next one's writes.)
        char *tp = lds + 3 * BUF + wave * kFTP;
        if constexpr (n == 9) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f_f32x4 *>(tp + l31 * 144 + 32 * j + 16 * lhi) =
                f_f32x4{pv[2 * j].x, pv[2 * j].y, pv[2 * j + 1].x, pv[2 * j + 1].y};
        } else if constexpr (n == 10) {
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) yrow[k2] = *reinterpret_cast<const f_f32x4 *>(tp + (8 * k2 + (lane >> 3)) * 144 + (lane & 7) * 16);
        } else {
          store_rows(0);
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) yrow[k2] = *reinterpret_cast<const f_f32x4 *>(tp + (8 * (k2 + 2) + (lane >> 3)) * 144 + (lane & 7) * 16);
        }
      }
      const f_f32x2 sh2 = {tsh, tsh};
#pragma unroll
      for (int j = 3 * (n - 9); j < (n == 11 ? 8 : 3 * (n - 8)); ++j) {
        const f_f32x2 d = pv[j] - sh2;
        t1 += d;
        t2 = __builtin_elementwise_fma(d, d, t2);
      }
    } else if constexpr (n == 12) {
      if constexpr (STORE) store_rows(2);
      // channels with a negative gamma are scanned negated (largest of -y)
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = pv[j] * sgn;
    } else if constexpr (n == 13) {
#pragma unroll
      for (int gq = 0; gq < G; ++gq) {
        float b0 = -__builtin_inff();
#pragma unroll
        for (int j = gq * QG / 2; j < (gq + 1) * QG / 2; ++j) b0 = fmaxf(b0, fmaxf(pv[j].x, pv[j].y));
        best[gq] = b0;
        at[gq] = 64;
      }
    } else if constexpr (n == 14 || n == 15) {
      // the first sample that has the largest value: upper half of the registers, then the lower
#pragma unroll
      for (int gq = 0; gq < G; ++gq) {
        constexpr int HQ = QG / 2;
        const int q_hi = gq * QG + (n == 14 ? QG : HQ) - 1;
#pragma unroll
        for (int q = q_hi; q > q_hi - HQ; --q) {
          const int nn = (q & 3) + 8 * ((q >> 2) - gq * (QG / 4)) + 4 * lhi;  // sample within the group
          const float val = (q & 1) ? pv[q >> 1].y : pv[q >> 1].x;
          at[gq] = val == best[gq] ? nn : at[gq];
        }
      }
    } else if constexpr (n == 16) {
#pragma unroll
      for (int gq = 0; gq < G; ++gq) {
        ob[gq] = __shfl_xor(best[gq], 32, kWave);
        oa[gq] = __shfl_xor(at[gq], 32, kWave);
      }
    } else if constexpr (n == 18) {
#pragma unroll
      for (int gq = 0; gq < G; ++gq) {
        float bb = best[gq];
        int aa = at[gq];
        if (ob[gq] > bb || (ob[gq] == bb && oa[gq] < aa)) { bb = ob[gq]; aa = oa[gq]; }
        if (aa == 64) aa = 0;  // (nothing compares: every sample a NaN)
        const float out = bb * sg;
        held[0] = held[1]; held[1] = held[2]; held[2] = held[3];
        held[3] = lhi == 0 ? __builtin_bit_cast(unsigned, out) : (unsigned)aa;
      }
    } else if constexpr (n == 19) {
      if (ce & 1) {
        // (mean, M2) of the lane's 32 samples of the tile, then of the channel's 64 (equal counts)
        const float s1 = t1.x + t1.y, s2 = t2.x + t2.y;
        mh = tsh + s1 * (1.0f / 32.0f);
        qh = fmaxf(s2 - s1 * s1 * (1.0f / 32.0f), 0.f);
        mo = __shfl_xor(mh, 32, kWave);
        qo = __shfl_xor(qh, 32, kWave);
      }
    } else if constexpr (n == 20) {
      if ((ce & 1) && real && lhi == 0) {
        const float dlt = mo - mh;
        const float mean = 0.5f * (mh + mo);
        const float m2w = (qh + qo) + 16.0f * dlt * dlt;  // n_a n_b / (n_a + n_b) = 16
        *reinterpret_cast<float2 *>(a.pairs + ((size_t)(ce >> 1) * kFM + ch) * 2) = make_float2(mean, m2w);
      }
    } else if constexpr (n == 21) {
      if ((ce & (4 / G - 1)) == 4 / G - 1 && real) {
        // four groups of this channel: 16 contiguous bytes per lane (values from the lower half-wave,
        // indices from the upper)
        const int b = ce / a.chunks_per_cloud;
        const int g_last = ((ce - b * a.chunks_per_cloud) * TN) / NS + G - 1;
        const size_t o = ((size_t)b * kFM + ch) * a.groups + g_last - 3 + (lhi ? a.ext_plane : 0);
        *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned *>(a.ext) + o) = make_uint4(held[0], held[1], held[2], held[3]);
      }
    }
  };
#define PF_PIECE(N) do { piece(std::integral_constant<int, (N)>{}); __builtin_amdgcn_sched_barrier(0); } while (0)

#pragma unroll
  for (int j = 0; j < 8; ++j) pv[j] = f_f32x2{0.f, 0.f};
  for (int c = c_lo; c < c_hi; ++c) {
    const int nxt = cur == 2 ? 0 : cur + 1;
    const char *Qc = lds + (size_t)cur * BUF;
    // buffer nxt held chunk c - 2: every wave has passed the barrier inside chunk c - 1, so all are
    // done with it.  (On the last chunk the clamped duplicate is staged: nobody reads it.)
    Qn = lds + (size_t)nxt * BUF;
    ce = c - 1;
    real = c > c_lo;
    // two accumulators, one per k-step parity: with vector instructions between them, MFMAs that
    // accumulate into ONE register block wait for one another (tools/micro/mfma_bf16_peak.py)
    f32x16 acc, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc1[q] = 0.f; }
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      const int ring = s & 1;
      // the next k-step's fragments -- behind the chunk's last step the next chunk's first (its image
      // is complete: the barrier below is behind us)
      if (s + 1 < K / 16) frag(Qc, s + 1, pf[ring ^ 1]);
      else frag(Qn, 0, pf[ring ^ 1]);
      __builtin_amdgcn_sched_barrier(0);  // the reads stay ahead of this step's MFMAs
      // T form: A = the chunk (rows of the block = its columns), B = W3 (columns = channels)
      const Split3 x = operand(pf[ring]);
      const Split3 &w = wsp[s];
#define PF_MM(AT, BT)                                                                                \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.AT, w.BT, acc, 0, 0, 0);
#define PF_MM1(AT, BT)                                                                               \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.AT, w.BT, acc1, 0, 0, 0);
#define PF_CASE(N) case N: PF_PIECE(N); break
      PF_MM(lo, hi) PF_MM1(hi, lo)
      switch (3 * s) {
        PF_CASE(0); PF_CASE(3); PF_CASE(6); PF_CASE(9); PF_CASE(12); PF_CASE(15); PF_CASE(18); PF_CASE(21);
      }
      PF_MM(mid, mid) PF_MM1(mid, hi)
      switch (3 * s + 1) {
        PF_CASE(1); PF_CASE(4); PF_CASE(7); PF_CASE(10); PF_CASE(13); PF_CASE(16); PF_CASE(19);
      }
      PF_MM(hi, mid) PF_MM1(hi, hi)
      switch (3 * s + 2) {
        PF_CASE(2); PF_CASE(5); PF_CASE(8); PF_CASE(11); PF_CASE(14); PF_CASE(20);
      }
#undef PF_CASE
#undef PF_MM
#undef PF_MM1
      if (s == 3) __syncthreads();  // every wave has staged chunk c + 1 and is past chunk c - 1
    }
    cur = nxt;
#pragma unroll
    for (int j = 0; j < 8; ++j) pv[j] = f_f32x2{acc[2 * j], acc[2 * j + 1]} + f_f32x2{acc1[2 * j], acc1[2 * j + 1]};
  }
  // the last chunk's epilogue
  ce = c_hi - 1;
  real = true;
  PF_PIECE(9); PF_PIECE(10); PF_PIECE(11); PF_PIECE(12); PF_PIECE(13); PF_PIECE(14); PF_PIECE(15);
  PF_PIECE(16); PF_PIECE(18); PF_PIECE(19); PF_PIECE(20); PF_PIECE(21);
#undef PF_PIECE
}

// ---- the same for 128 output channels (the 128 -> 128 layers of SA2 / SA3 / SA4, the vote aggregation
// and the IoU branch's grid MLP; its pooled last layers with nsample 16 / 32 / 64): FOUR waves per
// workgroup (wave = 32 channels), TWO workgroups per CU -- they are not in step with one another, so
// one's barrier and first-fragment round trip sit under the other's MFMAs --, two chunk buffers (three
// + the store tiles of two workgroups do not fit 160 KB): the barrier is at the top of a chunk and the
// chunk's first fragments are requested behind it.  A thread stages FOUR float4 per chunk (twice the
// vector work per MFMA of the 256-channel kernel); a slice is requested again as soon as its values are
// consumed, a chunk ahead of its use.  NS = 0: no pooling (statistics + y only); NS = 64: a group is two
// chunks.  DIRECT: the operand is x itself (no BatchNorm + ReLU in front).
struct Fwd128Args {
  int r, chunks_per_cloud, total_chunks, groups;
  const float *w;             // (128, 128)
  const float *x;             // (b, 128, r)
  const float *sc, *sh;       // (128) or null (DIRECT)
  const float *gamma;         // (128), pooled forms
  float *pairs;               // (b * r / 64, 128, 2)
  float *ext;                 // 2 planes of (b, 128, groups)
  size_t ext_plane;
  float *y;                   // (b, 128, r) or null
};

template <int NS, bool STORE, bool DIRECT>
__global__ void __launch_bounds__(256, 2) fwd128_kernel(const Fwd128Args a) {
  constexpr int M = 128, K = kFK, TN = kFTN, RP = kFRP, IMG = kFIMG, BUF = kFBUF;
  constexpr bool POOL = NS != 0;
  constexpr int G = NS == 16 ? 2 : 1;          // groups per chunk
  constexpr int CPG = NS == 64 ? 2 : 1;        // chunks per group
  constexpr int QG = 16 / G;                   // accumulator registers per group and chunk
  constexpr int FL = 4 * CPG / G;              // chunks per flush of four groups
  static_assert(NS == 0 || NS == 16 || NS == 32 || NS == 64, "nsample");
  extern __shared__ __attribute__((aligned(16))) char lds[];  // two chunk buffers + four store tiles

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int seg_row = tid >> 3, seg_c = (tid & 7) * 4;  // staging: rows seg_row + 32 it
  const int ch = 32 * wave + l31;

  Split3 wsp[K / 16];
  {
    // (the weight may sit at any 4-byte offset of the optimizer's flat parameter buffer)
    const float *wr = a.w + (size_t)ch * K + 8 * lhi;
    if ((reinterpret_cast<size_t>(a.w) & 15) == 0) {
#pragma unroll
      for (int s = 0; s < K / 16; ++s)
        wsp[s] = split3(*reinterpret_cast<const float4 *>(wr + 16 * s),
                        *reinterpret_cast<const float4 *>(wr + 16 * s + 4));
    } else {
#pragma unroll
      for (int s = 0; s < K / 16; ++s) {
        float w8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w8[j] = wr[16 * s + j];
        wsp[s] = split3(w8);
      }
    }
  }
  float sg = 1.f;
  if constexpr (POOL) sg = a.gamma[ch] < 0.f ? -1.f : 1.f;
  const f_f32x2 sgn = {sg, sg};
  float rsc[4], rsh[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    rsc[it] = DIRECT ? 1.f : a.sc[seg_row + 32 * it];
    rsh[it] = DIRECT ? 0.f : a.sh[seg_row + 32 * it];
  }

  // whole blocks of eight chunks per workgroup (pairs: two chunks; the extrema leave four groups at a time)
  const int blocks = a.total_chunks / 8;
  const int per = (blocks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int k_lo = (int)blockIdx.x * per;
  const int k_hi = k_lo + per < blocks ? k_lo + per : blocks;
  const int c_lo = 8 * k_lo, c_hi = 8 * k_hi;
  if (c_lo >= c_hi) return;

  float4 qx[4];
  auto fetch_slice = [&](int it, int c) {
    const int b = c / a.chunks_per_cloud;
    const int col0 = (c - b * a.chunks_per_cloud) * TN;
    qx[it] = *reinterpret_cast<const float4 *>(a.x + ((size_t)b * K + seg_row + 32 * it) * a.r + col0 + seg_c);
  };
  auto clampc = [&](int c) { return c < c_hi ? c : c_hi - 1; };

  const int tr_off = (8 * lhi + ((lane & 15) >> 2)) * RP + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  f_bf16x4 pf[2][3][2];
  auto frag = [&](const char *Q, int s, f_bf16x4 (&dst)[3][2]) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const char *p0 = Q + (size_t)t * IMG + (size_t)(16 * s) * RP + tr_off;
      dst[t][0] = f_lds_read_tr(p0);
      dst[t][1] = f_lds_read_tr(p0 + 4 * RP);
    }
  };
  auto operand = [&](const f_bf16x4 (&src)[3][2]) {
    Split3 sb;
    sb.hi = __builtin_shufflevector(src[0][0], src[0][1], 0, 1, 2, 3, 4, 5, 6, 7);
    sb.mid = __builtin_shufflevector(src[1][0], src[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
    sb.lo = __builtin_shufflevector(src[2][0], src[2][1], 0, 1, 2, 3, 4, 5, 6, 7);
    return sb;
  };

  float tsh = 0.f;
  f_f32x2 t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
  unsigned held[4] = {0u, 0u, 0u, 0u};
  float run_b = 0.f;  // NS = 64: the group's first chunk
  int run_a = 0;
  float sv[4], shh[4], sm[4], sl[4];
  f_f32x2 pv[8];
  float best[G], mh = 0.f, qh = 0.f, mo = 0.f, qo = 0.f, ob[G];
  int at[G], oa[G];
  char *Qn = lds;
  int ce = 0;
  bool real = false;
  f_f32x4 yrow[2];
  char *const tp = lds + 2 * BUF + wave * kFTP;
  auto store_rows = [&](int k0) {
    if (real) {
      const int b = ce / a.chunks_per_cloud;
      float *dst = a.y + ((size_t)b * M + 32 * wave + (lane >> 3)) * a.r + (ce - b * a.chunks_per_cloud) * TN + 4 * (lane & 7);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
        __builtin_nontemporal_store(yrow[k2], reinterpret_cast<f_f32x4 *>(dst + (size_t)(8 * (k0 + k2)) * a.r));
    }
  };
  // staging phases of slice `it` (A transform, B first term + the slice's next request, C second and third
  // terms, D images)
  auto stage_phase = [&](int it, int ph) {
    if (ph == 0) {
      const float xv[4] = {qx[it].x, qx[it].y, qx[it].z, qx[it].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) sv[e] = DIRECT ? xv[e] : fmaxf(__fmaf_rn(xv[e], rsc[it], rsh[it]), 0.f);
    } else if (ph == 1) {
      fetch_slice(it, clampc(ce + 3));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        shh[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sv[e]) & 0xffff0000u);
        sv[e] = sv[e] - shh[e];
      }
    } else if (ph == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sm[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sv[e]) & 0xffff0000u);
        sl[e] = sv[e] - sm[e];
      }
    } else {
      char *dst = Qn + (size_t)(seg_row + 32 * it) * RP + seg_c * 2;
      *reinterpret_cast<uint2 *>(dst) = make_uint2(pack_hi16(shh[0], shh[1]), pack_hi16(shh[2], shh[3]));
      *reinterpret_cast<uint2 *>(dst + IMG) = make_uint2(pack_hi16(sm[0], sm[1]), pack_hi16(sm[2], sm[3]));
      *reinterpret_cast<uint2 *>(dst + 2 * IMG) = make_uint2(pack_hi16(sl[0], sl[1]), pack_hi16(sl[2], sl[3]));
    }
  };
  // epilogue pieces of the previous chunk (numbered as in the 256-channel kernel)
  auto epi = [&](auto nt) {
    constexpr int n = decltype(nt)::value;
    if constexpr (n == 9 || n == 10 || n == 11) {
      if (n == 9 && (ce & 1) == 0) { tsh = pv[0].x; t1 = f_f32x2{0.f, 0.f}; t2 = t1; }
      if constexpr (STORE) {
        if constexpr (n == 9) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f_f32x4 *>(tp + l31 * 144 + 32 * j + 16 * lhi) =
                f_f32x4{pv[2 * j].x, pv[2 * j].y, pv[2 * j + 1].x, pv[2 * j + 1].y};
        } else if constexpr (n == 10) {
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) yrow[k2] = *reinterpret_cast<const f_f32x4 *>(tp + (8 * k2 + (lane >> 3)) * 144 + (lane & 7) * 16);
        } else {
          store_rows(0);
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) yrow[k2] = *reinterpret_cast<const f_f32x4 *>(tp + (8 * (k2 + 2) + (lane >> 3)) * 144 + (lane & 7) * 16);
        }
      }
      const f_f32x2 sh2 = {tsh, tsh};
#pragma unroll
      for (int j = 3 * (n - 9); j < (n == 11 ? 8 : 3 * (n - 8)); ++j) {
        const f_f32x2 d = pv[j] - sh2;
        t1 += d;
        t2 = __builtin_elementwise_fma(d, d, t2);
      }
    } else if constexpr (n == 12) {
      if constexpr (STORE) store_rows(2);
      if constexpr (POOL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = pv[j] * sgn;
      }
    } else if constexpr (n == 13) {
      if constexpr (POOL) {
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
          float b0 = -__builtin_inff();
#pragma unroll
          for (int j = gq * QG / 2; j < (gq + 1) * QG / 2; ++j) b0 = fmaxf(b0, fmaxf(pv[j].x, pv[j].y));
          best[gq] = b0;
          at[gq] = 64;
        }
      }
    } else if constexpr (n == 14 || n == 15) {
      if constexpr (POOL) {
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
          constexpr int HQ = QG / 2;
          const int q_hi = gq * QG + (n == 14 ? QG : HQ) - 1;
#pragma unroll
          for (int q = q_hi; q > q_hi - HQ; --q) {
            const int nn = (q & 3) + 8 * ((q >> 2) - gq * (QG / 4)) + 4 * lhi;
            const float val = (q & 1) ? pv[q >> 1].y : pv[q >> 1].x;
            at[gq] = val == best[gq] ? nn : at[gq];
          }
        }
      }
    } else if constexpr (n == 16) {
      if constexpr (POOL) {
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
          ob[gq] = __shfl_xor(best[gq], 32, kWave);
          oa[gq] = __shfl_xor(at[gq], 32, kWave);
        }
      }
    } else if constexpr (n == 18) {
      if constexpr (POOL) {
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
          float bb = best[gq];
          int aa = at[gq];
          if (ob[gq] > bb || (ob[gq] == bb && oa[gq] < aa)) { bb = ob[gq]; aa = oa[gq]; }
          if (aa == 64) aa = 0;
          bool push = true;
          if constexpr (CPG == 2) {
            if ((ce & 1) == 0) { run_b = bb; run_a = aa; push = false; }
            else if (bb > run_b) { aa += 32; }                 // the group's second chunk wins
            else { bb = run_b; aa = run_a; }                   // (ties: the first chunk's sample)
          }
          if (push) {
            const float out = bb * sg;
            held[0] = held[1]; held[1] = held[2]; held[2] = held[3];
            held[3] = lhi == 0 ? __builtin_bit_cast(unsigned, out) : (unsigned)aa;
          }
        }
      }
    } else if constexpr (n == 19) {
      if (ce & 1) {
        const float s1 = t1.x + t1.y, s2 = t2.x + t2.y;
        mh = tsh + s1 * (1.0f / 32.0f);
        qh = fmaxf(s2 - s1 * s1 * (1.0f / 32.0f), 0.f);
        mo = __shfl_xor(mh, 32, kWave);
        qo = __shfl_xor(qh, 32, kWave);
      }
    } else if constexpr (n == 20) {
      if ((ce & 1) && real && lhi == 0) {
        const float dlt = mo - mh;
        const float mean = 0.5f * (mh + mo);
        const float m2w = (qh + qo) + 16.0f * dlt * dlt;
        *reinterpret_cast<float2 *>(a.pairs + ((size_t)(ce >> 1) * M + ch) * 2) = make_float2(mean, m2w);
      }
    } else if constexpr (n == 21) {
      if constexpr (POOL) {
        if ((ce & (FL - 1)) == FL - 1 && real) {
          const int b = ce / a.chunks_per_cloud;
          const int g_last = ((ce - b * a.chunks_per_cloud) * TN) / NS + G - 1;
          const size_t o = ((size_t)b * M + ch) * a.groups + g_last - 3 + (lhi ? a.ext_plane : 0);
          *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned *>(a.ext) + o) = make_uint4(held[0], held[1], held[2], held[3]);
        }
      }
    }
  };
  using I9 = std::integral_constant<int, 9>;   using I10 = std::integral_constant<int, 10>;
  using I11 = std::integral_constant<int, 11>; using I12 = std::integral_constant<int, 12>;
  using I13 = std::integral_constant<int, 13>; using I14 = std::integral_constant<int, 14>;
  using I15 = std::integral_constant<int, 15>; using I16 = std::integral_constant<int, 16>;
  using I18 = std::integral_constant<int, 18>; using I19 = std::integral_constant<int, 19>;
  using I20 = std::integral_constant<int, 20>; using I21 = std::integral_constant<int, 21>;
  // the 24 slots behind the pairs of MFMAs: 0-15 the next chunk's four slices, 16-23 the previous
  // chunk's epilogue
  auto slot = [&](int n) {
    if (n < 16) stage_phase(n >> 2, n & 3);
    else if (n == 16) epi(I9{});
    else if (n == 17) epi(I10{});
    else if (n == 18) epi(I11{});
    else if (n == 19) { epi(I12{}); epi(I13{}); }
    else if (n == 20) { epi(I14{}); epi(I15{}); }
    else if (n == 21) epi(I16{});
    else if (n == 22) { epi(I18{}); epi(I19{}); }
    else { epi(I20{}); epi(I21{}); }
    __builtin_amdgcn_sched_barrier(0);
  };

  // run-in: chunk c_lo staged into buffer 0, chunk c_lo + 1 requested
  ce = c_lo - 2;  // (stage_phase requests chunk ce + 3)
#pragma unroll
  for (int it = 0; it < 4; ++it) fetch_slice(it, c_lo);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    stage_phase(it, 0); stage_phase(it, 1); stage_phase(it, 2); stage_phase(it, 3);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) pv[j] = f_f32x2{0.f, 0.f};

  for (int c = c_lo; c < c_hi; ++c) {
    __syncthreads();  // chunk c is staged by everyone, everyone is done with chunk c - 1
    const char *Qc = lds + (size_t)((c - c_lo) & 1) * BUF;
    Qn = lds + (size_t)(((c - c_lo) & 1) ^ 1) * BUF;
    ce = c - 1;
    real = c > c_lo;
    frag(Qc, 0, pf[0]);
    // (two accumulators where the registers allow it: the pooled forms keep one)
    f32x16 acc, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc1[q] = 0.f; }
#define F128_MM1(AT, BT)                                                                             \
  if constexpr (POOL) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.AT, w.BT, acc, 0, 0, 0);      \
  else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.AT, w.BT, acc1, 0, 0, 0)
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      const int ring = s & 1;
      if (s + 1 < K / 16) frag(Qc, s + 1, pf[ring ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      const Split3 x = operand(pf[ring]);
      const Split3 &w = wsp[s];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.lo, w.hi, acc, 0, 0, 0);
      F128_MM1(hi, lo);
      slot(3 * s);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.mid, w.mid, acc, 0, 0, 0);
      F128_MM1(mid, hi);
      slot(3 * s + 1);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.hi, w.mid, acc, 0, 0, 0);
      F128_MM1(hi, hi);
      slot(3 * s + 2);
    }
#undef F128_MM1
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pv[j] = f_f32x2{acc[2 * j], acc[2 * j + 1]};
      if constexpr (!POOL) pv[j] += f_f32x2{acc1[2 * j], acc1[2 * j + 1]};
    }
  }
  ce = c_hi - 1;
  real = true;
#pragma unroll
  for (int n = 16; n < 24; ++n) slot(n);
}

int pool_fwd256_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}

}  // namespace

// 1 when mlp_gemm_forward_stats_pool runs here: (m, k) = (256, 128), nsample 16 / 32, whole
// 256-column blocks per cloud (MLP_POOL_FWD256=0: the tiled kernel)
int mlp_pool_fwd256_supported(int b, int m, int k, int r, int ns, const float *w, const float *x) {
  const char *env = getenv("MLP_POOL_FWD256");  // (read on every call: the tests compare both kernels)
  const bool off = env && atoi(env) == 0;
  if (off || b <= 0 || m != kFM || k != kFK || r <= 0 || r % 256 != 0) return 0;
  if ((ns != 16 && ns != 32) || r % ns != 0) return 0;
  return ((reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(x)) & 15) == 0 ? 1 : 0;
}

// pairs: (b * r / 64, 256, 2); ext: 2 planes of (b, 256, r / ns) -- as the tiled kernel leaves them
int mlp_pool_fwd256_launch(int b, int r, int ns, const float *w, const float *x, const float *scale,
                           const float *shift, const float *gamma, float *y, float *pairs, float *ext,
                           hipStream_t stream) {
  if (!w || !x || !scale || !shift || !gamma || !pairs || !ext || (ns != 16 && ns != 32) || b <= 0 || r <= 0 ||
      r % 256 != 0)
    return (int)hipErrorInvalidValue;
  PoolFwdArgs a = {};
  a.r = r; a.chunks_per_cloud = r / kFTN; a.total_tiles = b * (r / 64); a.groups = r / ns;
  a.w = w; a.x = x; a.sc = scale; a.sh = shift; a.gamma = gamma; a.pairs = pairs; a.ext = ext; a.y = y;
  a.ext_plane = (size_t)b * kFM * (size_t)(r / ns);
  int grid = pool_fwd256_cus();
  if (grid > a.total_tiles / 2) grid = a.total_tiles / 2;
  const size_t kLds = 3 * (size_t)kFBUF + (y != nullptr ? 8 * (size_t)kFTP : 0);
  constexpr size_t kLdsMax = 3 * (size_t)kFBUF + 8 * (size_t)kFTP;
  static std::mutex mu;
  static bool attr_set = false;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pool_fwd256_kernel<16, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pool_fwd256_kernel<32, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pool_fwd256_kernel<16, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pool_fwd256_kernel<32, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
      attr_set = true;
    }
  }
  if (y != nullptr) {
    if (ns == 16) hipLaunchKernelGGL((pool_fwd256_kernel<16, true>), dim3(grid), dim3(512), kLds, stream, a);
    else hipLaunchKernelGGL((pool_fwd256_kernel<32, true>), dim3(grid), dim3(512), kLds, stream, a);
  } else {
    if (ns == 16) hipLaunchKernelGGL((pool_fwd256_kernel<16, false>), dim3(grid), dim3(512), kLds, stream, a);
    else hipLaunchKernelGGL((pool_fwd256_kernel<32, false>), dim3(grid), dim3(512), kLds, stream, a);
  }
  return pn2_launch_status();
}

// 1 when the (128, 128) layers' forward-with-statistics runs as fwd128_kernel: then
// mlp_gemm_forward_stats_parts hands out 64-column pairs for this shape and BOTH entries
// (mlp_gemm_forward_stats, mlp_gemm_forward_stats_pool) come here
int mlp_fwd128_enabled_for(int b, int m, int k, int r) {
  // OPT-IN (MLP_FWD128=1; read on every call: the tests compare both kernels).  Measured, B = 8: alone the
  // kernel wins from 2048 chunks on (8192 chunks 84 -> 70 us, input from HBM 95 -> 86; the pooled
  // nsample-64 layer of the IoU branch 97 -> 75; 2048 chunks 35 -> 33; 1024 chunks 28 -> 30), and inside the
  // profiled step its four launches are 36 us shorter than the tiled kernels' -- but the train step is
  // 0.12 ms SLOWER with it (6.03 against 5.91 ms, three runs each on one box; SUN RGB-D 9.42 against
  // 9.17; equal when the index chain does not run beside the main stream): two 68-KB workgroups per CU
  // leave the sampling kernels of the side stream no room, and the kernels behind it run at lower clocks.
  // The 256-channel kernel (one workgroup per CU) does not show this.
  const char *env = getenv("MLP_FWD128");
  if (!env || atoi(env) != 1) return 0;
  return b > 0 && m == 128 && k == kFK && r > 0 && r % 256 == 0 && (long long)b * r >= 65536 ? 1 : 0;
}

int mlp_fwd128_launch(int b, int r, int ns, int direct, const float *w, const float *x, const float *scale,
                      const float *shift, const float *gamma, float *y, float *pairs, float *ext,
                      hipStream_t stream) {
  if (!w || !x || !pairs || (!direct && (!scale || !shift)) || (ns != 0 && (!gamma || !ext)) || (ns == 0 && !y) ||
      (direct && ns != 0) || ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y)) & 15) ||
      (reinterpret_cast<size_t>(w) & 3) ||
      (ns != 0 && ns != 16 && ns != 32 && ns != 64) || (ns != 0 && r % ns != 0))
    return (int)hipErrorInvalidValue;
  Fwd128Args a = {};
  a.r = r; a.chunks_per_cloud = r / kFTN; a.total_chunks = b * (r / kFTN); a.groups = ns ? r / ns : 0;
  a.w = w; a.x = x; a.sc = scale; a.sh = shift; a.gamma = gamma; a.pairs = pairs; a.ext = ext; a.y = y;
  a.ext_plane = ns ? (size_t)b * 128 * (size_t)(r / ns) : 0;
  int grid = 2 * pool_fwd256_cus();
  {
    const char *env = getenv("MLP_FWD128_BLOCKS_PER_WG");
    const int bpw = env ? atoi(env) : 0;
    if (bpw > 0) grid = (a.total_chunks / 8 + bpw - 1) / bpw;
  }
  if (grid > a.total_chunks / 8) grid = a.total_chunks / 8;
  constexpr size_t kLds = 2 * (size_t)kFBUF + 4 * (size_t)kFTP;
  static std::mutex mu;
  static bool attr_set = false;
#define F128_ALL(X) X(0, true, false) X(0, true, true) X(16, true, false) X(16, false, false) X(32, true, false) \
  X(32, false, false) X(64, true, false) X(64, false, false)
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set) {
#define F128_ATTR(NS_, ST_, DI_)                                                                      \
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fwd128_kernel<NS_, ST_, DI_>),            \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
      F128_ALL(F128_ATTR)
#undef F128_ATTR
      attr_set = true;
    }
  }
  const bool st = y != nullptr, di = direct != 0;
#define F128_GO(NS_, ST_, DI_)                                                                        \
  if (ns == NS_ && st == ST_ && di == DI_)                                                            \
    hipLaunchKernelGGL((fwd128_kernel<NS_, ST_, DI_>), dim3(grid), dim3(256), kLds, stream, a);
  F128_ALL(F128_GO)
#undef F128_GO
#undef F128_ALL
  return pn2_launch_status();
}
