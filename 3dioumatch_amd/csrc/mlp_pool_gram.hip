// 3dioumatch_amd/csrc/mlp_pool_gram.hip -- backward of a max-pooled LAST shared-MLP layer (SA1:
// 64 -> 128 channels, pytorch_utils.py:14-39,70-124 + the max over nsample of
// pointnet2_modules.py:256-262) WITHOUT the layer's raw output y3 (gfx950).
//
// The incoming gradient of that layer is dy3 = a (g - c1 - xhat c2): the BatchNorm backward of a
// gradient g that is non-zero at ONE column per (channel, group) -- the pooled winner.  With
// xhat = (y3 - mu) is it splits into a dense affine part and a sparse part,
//     dy3 = q y3 + p + S,      q = -a is c2,  p = a (is c2 mu - c1),  S[c][s*(c,g)] = a dpooled[c][g]
// and y3 = W3 a2 (a2 = relu(bn(y2)), the layer's input).  Both products of the backward pass then
// need y3 only through the SMALL matrices
//     da2 = W3^T dy3 = (W3^T diag(q) W3) a2 + W3^T p + W3^T S        = M3 a2 + v + sparse
//     dW3 = dy3 a2^T = diag(q) W3 (a2 a2^T) + p (sum a2)^T + S a2^T  = diag(q) W3 C2 + p s2^T + R
// M3 (64 x 64) and v come from a tiny kernel before the pass; C2 = a2 a2^T (the 64 x 64 Gram matrix
// of the layer's input), s2 and R = S a2^T are accumulated by the pass; the sparse gradient S is a
// bf16 image in LDS that holds one entry per (channel, group) and rides the matrix pipe like the
// dense operands (first versions applied its 128 vector updates per group with LDS atomics --
// 1.3 ms per pass -- and with per-entry scalar loops -- 0.4 ms).  So the pass
// reads y2 once (268 MB at SA1) and writes da2 once -- y3 (537 MB) is neither read here NOR
// STORED BY THE FORWARD (csrc/mlp_chain.hip leaves only its statistics and pooled extrema): the
// largest activation of the network never exists in memory.  Matrix work per column:
// 2 x 64 x 64 (M3 a2, Gram) instead of 2 x 128 x 64 x 2.
//
// Shape of the pass: the bf16-split kernel of mlp_bwd_x6.h (operands split ONCE at staging into
// row-major bf16 images, two buffers, next chunk staged between the MFMA groups, one barrier per
// chunk); 32-column chunks; roles of equal matrix time, see the kernel.
#include "common.h"
#include "mlp_operand.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

typedef short g_bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ g_bf16x4 g_lds_read_tr(const char *p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) g_bf16x4 *)(__attribute__((address_space(3))) char *)p);
}

constexpr int kGM = 128, kGK = 64;  // the layer: kGM outputs, kGK inputs

// ---- before the pass: q, p per output channel; M3 = W3^T diag(q) W3; v = W3^T p
__global__ void __launch_bounds__(256)
pool_gram_prep_kernel(const float *__restrict__ w3, const float *__restrict__ coef3,
                      const float *__restrict__ mean3, const float *__restrict__ invstd3,
                      float *__restrict__ qp, float *__restrict__ m3, float *__restrict__ v) {
  __shared__ float q[kGM], p[kGM];
  const int tid = threadIdx.x;
  if (tid < kGM) {
    const float a = coef3[tid * 3], c1 = coef3[tid * 3 + 1], c2 = coef3[tid * 3 + 2];
    const float t = invstd3[tid] * c2;
    q[tid] = -(a * t);
    p[tid] = a * (t * mean3[tid] - c1);
    if (blockIdx.x == 0) { qp[tid * 2] = q[tid]; qp[tid * 2 + 1] = p[tid]; }
  }
  __syncthreads();
  const int e = blockIdx.x * 256 + tid;  // entry (k, k') of M3
  const int k = e >> 6, k2 = e & 63;
  double acc = 0.0;
  for (int c = 0; c < kGM; ++c) acc += (double)q[c] * (double)w3[c * kGK + k] * (double)w3[c * kGK + k2];
  m3[e] = (float)acc;
  if (blockIdx.x == 0 && tid < kGK) {
    double s = 0.0;
    for (int c = 0; c < kGM; ++c) s += (double)p[c] * (double)w3[c * kGK + tid];
    v[tid] = (float)s;
  }
}

struct GramArgs {
  int r, total_chunks, chunks_per_cloud, ns, groups;
  const float *y2;                       // (b, 64, r) raw output of the layer below
  const float *sc2, *sh2, *mean2, *invstd2;
  const float *m3, *v;                   // (64, 64), (64)
  const float *w3;                       // (128, 64)
  const float *coef3, *sc3, *sh3;        // (128, 3): a = coef3[3 c]; (128) each
  const int *argmax;                     // (b, 128, groups)
  const float *dpooled, *ymax;           // (b, 128, groups)
  float *dq;                             // (b, 64, r)
  float *part_c2, *part_s2, *part_r;     // per workgroup: 4096, 64, 8192 floats
  float *stats_part;                     // (64, workgroups, 2)
};

// The pass.  LDS per buffer: S images (the SPARSE gradient: three bf16 terms of a dpooled at
// [channel][winner's column], zero elsewhere -- every (channel, group) thread writes its entry when
// the chunk is staged and clears it after the chunk's MFMAs), a2 images, a raw copy of y2.
//   waves 0 / 1: da2 rows 32 w ..  = M3 a2 (4 steps) + W3^T S (8 steps) + v   -> 72 MFMAs
//   waves 2 / 3: R blocks S a2^T (rows {0,1} / {2,3} x both column blocks) and Gram blocks
//                a2 a2^T (row w - 2 x both column blocks)                       -> 72 MFMAs
// One barrier per chunk; the sparse parts ride the matrix pipe (a one-hot operand is exact).
__global__ void __launch_bounds__(256, 1) pool_gram_bwd_kernel(const GramArgs a) {
  constexpr int M = kGM, K = kGK, TN = 32;
  constexpr int RP = TN * 2 + 16;          // image row pitch, bytes
  constexpr int SIMG = M * RP, QIMG = K * RP;
  constexpr int RAWP = (TN + 4) * 4;       // raw copy row pitch, bytes
  constexpr int BUF = 3 * (SIMG + QIMG) + K * RAWP;
  constexpr int RCOFF = 2 * BUF, VOFF = RCOFF + K * 16;  // per row of a2: (sc, sh, mu, is); v
  constexpr int NG = 12;                   // MFMA groups per chunk of either role
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int seg_row = tid >> 3, seg_c = (tid & 7) * 4;
  const bool dgrad_wave = wave < 2;
  const int G = a.ns >= TN ? 1 : TN / a.ns;  // pooling groups that overlap a chunk

  // ---- once per workgroup: the S images start empty
  for (int t = tid; t < 2 * BUF / 16; t += 256) reinterpret_cast<uint4 *>(lds)[t] = make_uint4(0u, 0u, 0u, 0u);
  if (tid < K) {
    reinterpret_cast<float4 *>(lds + RCOFF)[tid] = make_float4(a.sc2[tid], a.sh2[tid], a.mean2[tid], a.invstd2[tid]);
    // (v in LDS, not in 16 registers of every lane: the kernel's two roles share one register budget,
    // and with it the weight-side waves spilled fragments to scratch INSIDE the chunk loop)
    reinterpret_cast<float *>(lds + VOFF)[tid] = a.v[tid];
  }
  RowCoef qc[2];
  size_t q_lane[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    qc[q] = {a.sc2[seg_row + 32 * q], a.sh2[seg_row + 32 * q], 0.f, 0.f, 0.f};
    q_lane[q] = (size_t)(seg_row + 32 * q) * a.r + seg_c;
  }
  // this thread's (channel, overlapping group) of the sparse gradient
  const int lc3 = tid & (M - 1), lgi = tid >> 7;
  const float l_sc3 = a.sc3[lc3], l_sh3 = a.sh3[lc3], l_a3 = a.coef3[lc3 * 3];
  int s_written[2] = {TN, TN};  // the column this thread's entry occupies in buffer 0 / 1 (TN: none)

  // waves 0 / 1: fragments of M3 (step s: k' = 16 s + 8 lhi + 0..7) and of W3^T (step s: channels
  // 16 s + 8 lhi + 0..7) for da2 rows 32 wave + l31, split once; v
  Split3 msp[K / 16], wsp[M / 16];
  if (dgrad_wave) {
    const float *mr = a.m3 + (size_t)(32 * wave + l31) * K;
#pragma unroll
    for (int s = 0; s < K / 16; ++s)
      msp[s] = split3(*reinterpret_cast<const float4 *>(mr + 16 * s + 8 * lhi),
                      *reinterpret_cast<const float4 *>(mr + 16 * s + 8 * lhi + 4));
    const float *wc = a.w3 + 32 * wave + l31;
#pragma unroll
    for (int s = 0; s < M / 16; ++s) {
      float w8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w8[j] = wc[(size_t)(16 * s + 8 * lhi + j) * K];
      wsp[s] = split3(w8);
    }
  }
  // waves 2 / 3: R blocks (row block 2 (wave - 2) + i, column block j), Gram blocks (wave - 2, j)
  // (the two roles' accumulators that live across all chunks share registers: the weight-side waves'
  // six blocks, of which the data-side waves use two as their BatchNorm sums -- declared apart they
  // cost 32 registers more and the kernel spilled fragments inside the chunk loop)
  f32x16 accX[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) accX[j][q] = 0.f;
  f32x16 (&accR)[2][2] = *reinterpret_cast<f32x16 (*)[2][2]>(&accX[0]);
  f32x16 (&accC)[2] = *reinterpret_cast<f32x16 (*)[2]>(&accX[4]);
  f32x16 &st1 = accX[0], &st2 = accX[1];
  float s2acc[2] = {0.f, 0.f};

  const int per = (a.total_chunks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = (int)blockIdx.x * per;
  const int c_hi = c_lo + per < a.total_chunks ? c_lo + per : a.total_chunks;

  // raw operands of one chunk in registers
  float4 qx[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  int l_am = 0, l_base = 0;
  float l_dp = 0.f, l_ym = 0.f;

  auto fetch_item = [&](int item, int c) {
    const int b = c / a.chunks_per_cloud;
    const int col0 = (c - b * a.chunks_per_cloud) * TN;
    if (item < 2) {
      qx[item] = *reinterpret_cast<const float4 *>(a.y2 + (size_t)b * K * a.r + col0 + q_lane[item]);
    } else if (lgi < G) {
      const int g = col0 / a.ns + lgi;
      const size_t gi = ((size_t)b * M + lc3) * a.groups + g;
      l_am = a.argmax[gi];
      l_dp = a.dpooled[gi];
      l_ym = a.ymax[gi];
      l_base = g * a.ns - col0;
    }
  };
  // the winner's column in the chunk gets a dpooled -- unless it lies outside the chunk or the ReLU
  // behind the pool was shut: then the entry goes to column 32, the rows' padding, which no fragment
  // read touches (no branch: the piece sits between two MFMAs)
  auto stage_sparse = [&](int buf, bool real) {
    const int sl = l_base + l_am;
    const bool hit = lgi < G && real && __fmaf_rn(l_ym, l_sc3, l_sh3) > 0.f && sl >= 0 && sl < TN;
    const int s = hit ? sl : TN;
    s_written[buf] = s;
    const float adp = l_a3 * l_dp;
    const float hf = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, adp) & 0xffff0000u);
    const float r1 = adp - hf;
    const float mf = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
    const float lf = r1 - mf;
    unsigned short *dst = reinterpret_cast<unsigned short *>(lds + (size_t)buf * BUF + (size_t)lc3 * RP) + s;
    dst[0] = (unsigned short)(__builtin_bit_cast(unsigned, hf) >> 16);
    dst[SIMG / 2] = (unsigned short)(__builtin_bit_cast(unsigned, mf) >> 16);
    dst[SIMG] = (unsigned short)(__builtin_bit_cast(unsigned, lf) >> 16);
  };
  // (real: the chunk exists -- the pipeline re-stages the last chunk past the end, never consumed)
  auto stage_item = [&](int item, int buf, bool real) {
    char *base = lds + (size_t)buf * BUF;
    if (item < 2) {
      const int row = seg_row + 32 * item;
      const float xv[4] = {qx[item].x, qx[item].y, qx[item].z, qx[item].w};
      float v[4], h[4], m[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = transform<OP_BNRELU>(xv[e], 0.f, qc[item]);
        h[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[e]) & 0xffff0000u);
        const float r1 = v[e] - h[e];
        m[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
        l[e] = r1 - m[e];
      }
      if (real) s2acc[item] += (v[0] + v[1]) + (v[2] + v[3]);
      char *dst = base + 3 * SIMG + (size_t)row * RP + seg_c * 2;
      *reinterpret_cast<uint2 *>(dst) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
      *reinterpret_cast<uint2 *>(dst + QIMG) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
      *reinterpret_cast<uint2 *>(dst + 2 * QIMG) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
      *reinterpret_cast<float4 *>(base + 3 * (SIMG + QIMG) + (size_t)row * RAWP + seg_c * 4) = qx[item];
    } else {
      stage_sparse(buf, real);
    }
  };
  // the S image is empty again after the chunk's MFMAs: this thread clears its entry
  auto clear_entry = [&](int buf) {
    unsigned short *dst = reinterpret_cast<unsigned short *>(lds + (size_t)buf * BUF + (size_t)lc3 * RP) + s_written[buf];
    dst[0] = 0; dst[SIMG / 2] = 0; dst[SIMG] = 0;
  };
  auto clampc = [&](int c) { return c < c_hi ? c : c_hi - 1; };
  __syncthreads();  // the zero fill is complete
  if (c_lo < c_hi) {
#pragma unroll
    for (int it = 0; it < 3; ++it) fetch_item(it, c_lo);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      stage_item(it, 0, true);
      fetch_item(it, clampc(c_lo + 1));
    }
  }
  __syncthreads();

  const int tr_off = (8 * lhi + ((lane & 15) >> 2)) * RP + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int rw_off = l31 * RP + 8 * lhi * 2;

  float pv[4], ph_[4], pm[4], pl[4];  // a y2 slice between the pieces of its staging
  // (the role is a compile-time argument and the loop over the chunks is written once per role: with
  // the role tested inside one common loop both roles' long-lived registers -- 144 of fragments here,
  // 96 of accumulators there -- were live across it, and the kernel spilled inside the loop)
  auto chunk = [&](auto curt, auto rolet, int c) {
    constexpr int cur = decltype(curt)::value;
    constexpr bool kDgrad = decltype(rolet)::value;
    const char *Sc = lds + (size_t)cur * BUF, *Qc = Sc + 3 * SIMG;
    const int ahead = clampc(c + 2);
    // The next chunk's staging in FOURTEEN pieces of a few vector instructions, one behind each of the
    // first pairs of MFMAs of either role (36 pairs per chunk), a scheduling barrier behind every piece:
    // left to itself the compiler issues a step's MFMAs back to back and the staging after them, where
    // (one wave per SIMD) nothing is in flight to hide it -- 125 of the kernel's 475 us at SA1.
    // Pieces 0-4 / 5-9: a y2 slice (transform; first term; second and third; images; raw copy + the
    // load of the chunk after); 10: this thread's entry of chunk c - 1 in buffer cur ^ 1 cleared (read
    // by everyone before the barrier that ended that chunk); 11: the next entry written; 12: its loads.
    const bool next_real = c + 1 < c_hi;
    auto piece = [&](int k) {
#if !defined(GRAM_ABL) || GRAM_ABL != 1   // (timing ablation 1: no staging / loads inside the loop)
      char *base = lds + (size_t)(cur ^ 1) * BUF;
      if (k < 10) {
        const int item = k / 5, ph = k % 5, row = seg_row + 32 * item;
        if (ph == 0) {
          const float xv[4] = {qx[item].x, qx[item].y, qx[item].z, qx[item].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) pv[e] = transform<OP_BNRELU>(xv[e], 0.f, qc[item]);
          if (next_real) s2acc[item] += (pv[0] + pv[1]) + (pv[2] + pv[3]);
        } else if (ph == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            ph_[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, pv[e]) & 0xffff0000u);
            pv[e] = pv[e] - ph_[e];
          }
        } else if (ph == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pm[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, pv[e]) & 0xffff0000u);
            pl[e] = pv[e] - pm[e];
          }
        } else if (ph == 3) {
          char *dst = base + 3 * SIMG + (size_t)row * RP + seg_c * 2;
          *reinterpret_cast<uint2 *>(dst) = make_uint2(pack_hi16(ph_[0], ph_[1]), pack_hi16(ph_[2], ph_[3]));
          *reinterpret_cast<uint2 *>(dst + QIMG) = make_uint2(pack_hi16(pm[0], pm[1]), pack_hi16(pm[2], pm[3]));
          *reinterpret_cast<uint2 *>(dst + 2 * QIMG) = make_uint2(pack_hi16(pl[0], pl[1]), pack_hi16(pl[2], pl[3]));
        } else {
          *reinterpret_cast<float4 *>(base + 3 * (SIMG + QIMG) + (size_t)row * RAWP + seg_c * 4) = qx[item];
          fetch_item(item, ahead);
        }
      } else if (k == 10) {
        clear_entry(cur ^ 1);
      } else if (k == 11) {
        stage_sparse(cur ^ 1, next_real);
      } else if (k == 12) {
        fetch_item(2, ahead);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    };
    const int b = c / a.chunks_per_cloud;
    const int col0 = (c - b * a.chunks_per_cloud) * TN;
    if constexpr (kDgrad) {
      f32x16 accD;
#pragma unroll
      for (int q = 0; q < 16; ++q) accD[q] = 0.f;
      g_bf16x4 pf[2][3][2];
      // transposing reads of an image: for this lane's column, eight consecutive rows from 16 s on
      auto frag = [&](const char *img, int term_bytes, int s, g_bf16x4 (&dst)[3][2]) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const char *p0 = img + (size_t)t * term_bytes + (size_t)(16 * s) * RP + tr_off;
          dst[t][0] = g_lds_read_tr(p0);
          dst[t][1] = g_lds_read_tr(p0 + 4 * RP);
        }
      };
      auto operand = [&](const g_bf16x4 (&src)[3][2]) {
        Split3 sb;
        sb.hi = __builtin_shufflevector(src[0][0], src[0][1], 0, 1, 2, 3, 4, 5, 6, 7);
        sb.mid = __builtin_shufflevector(src[1][0], src[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
        sb.lo = __builtin_shufflevector(src[2][0], src[2][1], 0, 1, 2, 3, 4, 5, 6, 7);
        return sb;
      };
      // dense part: M3 (registers) * a2; sparse part: W3^T (registers) * S.  Two accumulators, the
      // MFMAs of two consecutive steps interleaved: 72 MFMAs on ONE accumulator are a chain of
      // dependent-issue latencies and made these two waves the pole of every chunk
      f32x16 accE;
#pragma unroll
      for (int q = 0; q < 16; ++q) accE[q] = 0.f;
      auto step_operand = [&](int g) -> const Split3 & { return g < K / 16 ? msp[g] : wsp[g - K / 16]; };
      auto step_frag = [&](int g, g_bf16x4 (&dst)[3][2]) {
        if (g < K / 16) frag(Qc, QIMG, g, dst);
        else frag(Sc, SIMG, g - K / 16, dst);
      };
      g_bf16x4 pg[2][3][2];
      step_frag(0, pf[0]);
      step_frag(1, pg[0]);
#pragma unroll
      for (int g = 0; g < NG; g += 2) {
        if (g + 2 < NG) {
          step_frag(g + 2, pf[((g >> 1) + 1) & 1]);
          step_frag(g + 3, pg[((g >> 1) + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);  // (the fragment reads of the NEXT two steps stay up here)
        const Split3 b0 = operand(pf[(g >> 1) & 1]), b1 = operand(pg[(g >> 1) & 1]);
        const Split3 &a0 = step_operand(g), &a1 = step_operand(g + 1);
#define GR_STEP(AT, BT, SLOT)                                                                       \
  accD = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.AT, b0.BT, accD, 0, 0, 0);                      \
  accE = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.AT, b1.BT, accE, 0, 0, 0);                      \
  piece(6 * (g >> 1) + SLOT)
        GR_STEP(lo, hi, 0);
        GR_STEP(hi, lo, 1);
        GR_STEP(mid, mid, 2);
        GR_STEP(mid, hi, 3);
        GR_STEP(hi, mid, 4);
        GR_STEP(hi, hi, 5);
#undef GR_STEP
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) accD[q] += accE[q];
      float *dst = a.dq + ((size_t)b * K + 32 * wave + 4 * lhi) * a.r + col0 + l31;
      const float4 *rc = reinterpret_cast<const float4 *>(lds + RCOFF);
      const char *raw = Sc + 3 * (SIMG + QIMG);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        // this lane: column l31, row 32 wave + 4 lhi + (q & 3) + 8 (q >> 2)
        const int ro = (q & 3) + 8 * (q >> 2);
        const int row = 32 * wave + 4 * lhi + ro;
        const float d = accD[q] + reinterpret_cast<const float *>(lds + VOFF)[row];
#if !defined(GRAM_ABL) || GRAM_ABL != 2   // (timing ablation 2: no stores of the tile)
        __builtin_nontemporal_store(d, &dst[(size_t)ro * a.r]);
#endif
        const float4 c4 = rc[row];
        const float yv = *reinterpret_cast<const float *>(raw + (size_t)row * RAWP + l31 * 4);
        const float gg = __fmaf_rn(yv, c4.x, c4.y) > 0.f ? d : 0.f;
        st1[q] += gg;
        st2[q] = __fmaf_rn(gg, (yv - c4.z) * c4.w, st2[q]);
      }
    } else {
      // R blocks += S rows x a2 rows, Gram blocks += a2 rows x a2 rows (row-wise reads)
      const int i0 = wave - 2;
      auto rows = [&](const char *img, int term_bytes, int blk, int s) {
        Split3 f;
        const char *p0 = img + (size_t)(blk * 32) * RP + rw_off + 16 * s * 2;
        f.hi = *reinterpret_cast<const bf16x8 *>(p0);
        f.mid = *reinterpret_cast<const bf16x8 *>(p0 + term_bytes);
        f.lo = *reinterpret_cast<const bf16x8 *>(p0 + 2 * term_bytes);
        return f;
      };
      // six MFMAs of a block in three pairs, a staging piece behind each
#define X6_PAIRS(ACC, A, B, SLOT0)                                                                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).lo, (B).hi, ACC, 0, 0, 0);                      \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).hi, (B).lo, ACC, 0, 0, 0);                      \
  piece(SLOT0);                                                                                     \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).mid, (B).mid, ACC, 0, 0, 0);                    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).mid, (B).hi, ACC, 0, 0, 0);                     \
  piece(SLOT0 + 1);                                                                                 \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).hi, (B).mid, ACC, 0, 0, 0);                     \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).hi, (B).hi, ACC, 0, 0, 0);                      \
  piece(SLOT0 + 2)
#pragma unroll
      for (int s = 0; s < TN / 16; ++s) {
        Split3 sq[2], sp[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) sq[j] = rows(Qc, QIMG, j, s);
#pragma unroll
        for (int i = 0; i < 2; ++i) sp[i] = rows(Sc, SIMG, 2 * i0 + i, s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) { X6_PAIRS(accR[i][j], sp[i], sq[j], 18 * s + 6 * i + 3 * j); }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // (sq[i0] with the wave's number as the index put the fragments into scratch memory: six
          // 16-byte stores and three loads per step, each behind a wait for every outstanding load
          // and store of the wave -- a uniform branch instead)
          if (i0 == 0) { X6_PAIRS(accC[j], sq[0], sq[j], 18 * s + 12 + 3 * j); }
          else { X6_PAIRS(accC[j], sq[1], sq[j], 18 * s + 12 + 3 * j); }
        }
      }
#undef X6_PAIRS
    }
    __syncthreads();  // chunk c read by everyone, chunk c+1 staged by everyone
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  if (dgrad_wave) {
    for (int c = c_lo; c < c_hi; c += 2) {
      chunk(B0{}, std::true_type{}, c);
      if (c + 1 < c_hi) chunk(B1{}, std::true_type{}, c + 1);
    }
  } else {
    for (int c = c_lo; c < c_hi; c += 2) {
      chunk(B0{}, std::false_type{}, c);
      if (c + 1 < c_hi) chunk(B1{}, std::false_type{}, c + 1);
    }
  }

  // ---- per-workgroup partials
  if (dgrad_wave) {
    // the lanes' column sums -> row sums through LDS (the buffers are free: the loop's last barrier
    // is behind every wave, and each wave parks and reads only its own rows)
    const int parts = (int)gridDim.x;
    float2 *park = reinterpret_cast<float2 *>(lds) + (size_t)wave * 32 * 33;
#pragma unroll
    for (int q = 0; q < 16; ++q) park[(4 * lhi + (q & 3) + 8 * (q >> 2)) * 33 + l31] = make_float2(st1[q], st2[q]);
    if (lane < 32) {  // (same wave wrote: LDS operations complete in order)
      float a1 = 0.f, a2 = 0.f;
      for (int c2 = 0; c2 < 32; ++c2) { const float2 v = park[lane * 33 + c2]; a1 += v.x; a2 += v.y; }
      const int row = 32 * wave + lane;
      a.stats_part[((size_t)row * parts + blockIdx.x) * 2] = a1;
      a.stats_part[((size_t)row * parts + blockIdx.x) * 2 + 1] = a2;
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float s = s2acc[q];
    s += __shfl_xor(s, 1, kWave);
    s += __shfl_xor(s, 2, kWave);
    s += __shfl_xor(s, 4, kWave);
    if ((tid & 7) == 0) a.part_s2[(size_t)blockIdx.x * K + seg_row + 32 * q] = s;
  }
  if (!dgrad_wave) {
    const int i0 = wave - 2;
    float *oc = a.part_c2 + (size_t)blockIdx.x * K * K;
    float *orr = a.part_r + (size_t)blockIdx.x * M * K;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ro = (q & 3) + 8 * (q >> 2) + 4 * lhi;
        oc[(size_t)(32 * i0 + ro) * K + 32 * j + l31] = accC[j][q];
#pragma unroll
        for (int i = 0; i < 2; ++i) orr[(size_t)(32 * (2 * i0 + i) + ro) * K + 32 * j + l31] = accR[i][j][q];
      }
  }
}

// C2 (64 x 64), s2 (64) and R (128 x 64) from the workgroups' partials, in double: a workgroup sums
// 32 consecutive elements, its 8 slices of lanes each an eighth of the partials (four loads in flight)
constexpr int kGramSums = kGK * kGK + kGK + kGM * kGK;  // 12 352 = 386 x 32

__global__ void __launch_bounds__(256)
pool_gram_reduce_kernel(int parts, const float *__restrict__ part_c2, const float *__restrict__ part_s2,
                        const float *__restrict__ part_r, double *__restrict__ sums) {
  __shared__ double red[8][32];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
  const float *src;
  size_t stride;
  if (e < kGK * kGK) { src = part_c2 + e; stride = kGK * kGK; }
  else if (e < kGK * kGK + kGK) { src = part_s2 + (e - kGK * kGK); stride = kGK; }
  else { src = part_r + (e - kGK * kGK - kGK); stride = kGM * kGK; }
  double s = 0.0;
  int p = sl;
  for (; p + 24 < parts; p += 32) {
    const float v0 = src[(size_t)p * stride], v1 = src[(size_t)(p + 8) * stride];
    const float v2 = src[(size_t)(p + 16) * stride], v3 = src[(size_t)(p + 24) * stride];
    s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
  }
  for (; p < parts; p += 8) s += (double)src[(size_t)p * stride];
  red[sl][threadIdx.x & 31] = s;
  __syncthreads();
  if (sl == 0) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][threadIdx.x];
    sums[e] = t;
  }
}

// dW3[c][k] = q_c sum_k' W3[c][k'] C2[k'][k] + p_c s2[k] + R[c][k]
__global__ void __launch_bounds__(256)
pool_gram_dw_kernel(const float *__restrict__ w3, const float *__restrict__ qp,
                    const double *__restrict__ sums, float *__restrict__ dw) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= kGM * kGK) return;
  const int c = e >> 6, k = e & 63;
  const double *c2 = sums, *s2 = sums + kGK * kGK, *rr = s2 + kGK;
  double acc = 0.0;
  for (int k2 = 0; k2 < kGK; ++k2) acc += (double)w3[c * kGK + k2] * c2[k2 * kGK + k];
  dw[e] = (float)((double)qp[c * 2] * acc + (double)qp[c * 2 + 1] * s2[k] + rr[e]);
}

int gram_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}

int gram_workgroups(int b, int r) {
  const long long total = (long long)b * (r / 32);
  long long g = gram_cus();
  if (g > total / 8) g = total / 8;
  return (int)(g < 1 ? 1 : g);
}

constexpr size_t kGramLds = 2 * (3 * (128 + 64) * 80 + 64 * 144) + 64 * 16 + 64 * 4;

}  // namespace

#define MLP_API extern "C" __attribute__((visibility("default")))

// 1 when mlp_pool_gram_backward covers the layer: (m, k) = (128, 64), nsample 16 / 32 / 64, whole
// 32-column chunks per cloud
MLP_API int mlp_pool_gram_supported(int b, int m, int k, int r, int ns) {
  static const bool off = getenv("MLP_POOL_GRAM") && atoi(getenv("MLP_POOL_GRAM")) == 0;
  if (off || b <= 0 || m != kGM || k != kGK || r <= 0 || r % 32 != 0) return 0;
  if ((ns != 16 && ns != 32 && ns != 64) || r % ns != 0) return 0;
  return (long long)b * (r / 32) >= 64 ? 1 : 0;
}

// number of per-workgroup partials (= parts of stats_part) and floats of workspace
MLP_API int mlp_pool_gram_parts(int b, int r) { return gram_workgroups(b, r); }
MLP_API size_t mlp_pool_gram_workspace_floats(int b, int r) {
  const size_t g = (size_t)gram_workgroups(b, r);
  // qp (256) + M3 (4096) + v (64) + partials (4096 + 64 + 8192 per workgroup) + the sums as doubles
  return 256 + 4096 + 64 + g * (4096 + 64 + 8192) + 2 * (size_t)kGramSums + 16;
}

// Backward of the pooled last layer y3 = w3 . relu(bn2(y2)) from y2 and the pooled tensors alone:
// dq (b,64,r) = gradient w.r.t. relu(bn2(y2)); dw3 (128,64); stats_part (64, parts, 2): the
// BatchNorm-backward sums of layer 2 (for mlp_bn_backward_finalize).  coef3 (128,3) = (a, c1, c2) of
// layer 3 as mlp_bn_relu_pool_backward leaves them.
MLP_API int mlp_pool_gram_backward(int b, int r, int ns, const float *w3, const float *y2, const float *sc2,
                                   const float *sh2, const float *mean2, const float *invstd2,
                                   const float *coef3, const float *sc3, const float *sh3,
                                   const float *mean3, const float *invstd3, const int *argmax,
                                   const float *dpooled, const float *ymax, float *dq, float *dw3,
                                   float *stats_part, float *workspace, void *stream_) {
  if (!mlp_pool_gram_supported(b, kGM, kGK, r, ns) || !w3 || !y2 || !sc2 || !sh2 || !mean2 || !invstd2 ||
      !coef3 || !sc3 || !sh3 || !mean3 || !invstd3 || !argmax || !dpooled || !ymax || !dq || !dw3 ||
      !stats_part || !workspace || (reinterpret_cast<size_t>(workspace) & 15) ||
      (reinterpret_cast<size_t>(y2) & 15))
    return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  const int g = gram_workgroups(b, r);
  float *qp = workspace, *m3 = qp + 256, *v = m3 + 4096;
  float *part_c2 = v + 64, *part_s2 = part_c2 + (size_t)g * 4096, *part_r = part_s2 + (size_t)g * 64;
  float *tail = part_r + (size_t)g * 8192;
  double *sums = reinterpret_cast<double *>(tail + ((reinterpret_cast<size_t>(tail) & 7) ? 1 : 0));
  hipLaunchKernelGGL(pool_gram_prep_kernel, dim3(16), dim3(256), 0, stream, w3, coef3, mean3, invstd3, qp,
                     m3, v);
  static std::mutex mu;
  static bool attr_set = false;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pool_gram_bwd_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGramLds);
      attr_set = true;
    }
  }
  GramArgs a = {};
  a.r = r; a.total_chunks = b * (r / 32); a.chunks_per_cloud = r / 32; a.ns = ns; a.groups = r / ns;
  a.y2 = y2; a.sc2 = sc2; a.sh2 = sh2; a.mean2 = mean2; a.invstd2 = invstd2;
  a.m3 = m3; a.v = v; a.w3 = w3; a.coef3 = coef3; a.sc3 = sc3; a.sh3 = sh3;
  a.argmax = argmax; a.dpooled = dpooled; a.ymax = ymax;
  a.dq = dq; a.part_c2 = part_c2; a.part_s2 = part_s2; a.part_r = part_r; a.stats_part = stats_part;
  hipLaunchKernelGGL(pool_gram_bwd_kernel, dim3(g), dim3(256), kGramLds, stream, a);
  static_assert(kGramSums % 32 == 0, "whole reduce workgroups");
  hipLaunchKernelGGL(pool_gram_reduce_kernel, dim3(kGramSums / 32), dim3(256), 0, stream, g, part_c2, part_s2,
                     part_r, sums);
  hipLaunchKernelGGL(pool_gram_dw_kernel, dim3(kGM * kGK / 256), dim3(256), 0, stream, w3, qp, sums, dw3);
  return pn2_launch_status();
}
