// 3dioumatch_amd/csrc/mlp_pool_gram256.hip -- backward of the max-pooled LAST shared-MLP layer of
// SA2 / SA3 / SA4 (128 -> 256 channels, pytorch_utils.py:14-39,70-124 + the max over nsample of
// pointnet2_modules.py:256-262) WITHOUT the layer's raw output y3 (gfx950).
//
// The algebra is that of mlp_pool_gram.hip (SA1's 64 -> 128 layer): with dy3 = q y3 + p + S
// (S = one entry per (channel, group), the pooled winner) and y3 = W3 a2,
//     da2 = (W3^T diag(q) W3) a2 + W3^T p + W3^T S        = M3 a2 + v + sparse
//     dW3 = diag(q) W3 (a2 a2^T) + p (sum a2)^T + S a2^T  = diag(q) W3 C2 + p s2^T + R
// so the forward pass stores no (B,256,m,ns) tensor (268 MB at SA2) and this pass reads y2 instead of
// (y2, y3).  At 256 x 128 one workgroup cannot hold all the operands of the four products (W3^T and
// M3 as split register fragments are 288 registers per lane, the R and Gram accumulators 176 more),
// so the pass is TWO kernels over the same 32-column chunks of y2:
//   * pool_gram256_dgrad_kernel: da2 (+ the BatchNorm-backward sums of the layer below from the
//     tiles in the accumulators).  A wave owns 32 rows of da2; M3 and W3^T rows live in registers,
//     split once.  The sparse operand is an image S^T[n][c] in LDS (three bf16 terms): every
//     (channel, group) thread writes its entry at [winner's column][channel], the wave reads its
//     fragment (column n, eight consecutive channels) as ONE 16-byte read per term.  The image is
//     single-buffered: sparse steps first, a barrier, then the dense steps during which the entries
//     are cleared and the next chunk's written.
//   * pool_gram256_wgrad_kernel: R = S a2^T (a wave owns 64 channels x all 128 columns), the Gram
//     matrix C2 = a2 a2^T (its 10 upper blocks over the four waves) and s2.  Here a lane's one-hot
//     fragment of S (its channel, eight consecutive columns) is built in registers from the pooled
//     tensors it loads itself -- no image.
// Both read a2 as the three-term bf16 images of mlp_bwd_x6.h (split once at staging).
#include "common.h"
#include "mlp_operand.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

typedef short h_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned h_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ h_bf16x4 h_lds_read_tr(const char *p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) h_bf16x4 *)(__attribute__((address_space(3))) char *)p);
}

constexpr int kHM = 256, kHK = 128, kHTN = 32;
constexpr int kHRP = kHTN * 2 + 16;          // image row pitch (bytes): 80
constexpr int kHQIMG = kHK * kHRP;           // bytes per term of the a2 image
constexpr int kHRAWP = (kHTN + 4) * 4;       // raw copy row pitch
constexpr int kHRAW = kHK * kHRAWP;
constexpr int kHSP = kHM * 2 + 16;           // S^T row pitch (bytes): 528 = 33 x 16
constexpr int kHSIMG = kHTN * kHSP;          // bytes per term of S^T

// ---- before the pass: q, p per output channel; M3 = W3^T diag(q) W3 (128 x 128); v = W3^T p
__global__ void __launch_bounds__(256)
pool_gram256_prep_kernel(const float *__restrict__ w3, const float *__restrict__ coef3,
                         const float *__restrict__ mean3, const float *__restrict__ invstd3,
                         float *__restrict__ qp, float *__restrict__ m3, float *__restrict__ v) {
  // one workgroup per row k of M3; thread (k' = tid & 127, half = tid >> 7) sums half of the channels
  __shared__ float q[kHM], p[kHM];
  __shared__ double part[kHK];
  const int tid = threadIdx.x, k = blockIdx.x;
  {
    const float a = coef3[tid * 3], c1 = coef3[tid * 3 + 1], c2 = coef3[tid * 3 + 2];
    const float t = invstd3[tid] * c2;
    q[tid] = -(a * t);
    p[tid] = a * (t * mean3[tid] - c1);
    if (k == 0) { qp[tid * 2] = q[tid]; qp[tid * 2 + 1] = p[tid]; }
  }
  __syncthreads();
  const int k2 = tid & (kHK - 1), half = tid >> 7;
  double acc = 0.0, accv = 0.0;
  for (int c0 = half * (kHM / 2); c0 < (half + 1) * (kHM / 2); c0 += 16) {
    float wk[16], wk2[16];  // (sixteen rows' loads in flight: the loop is a chain of L2 round trips otherwise)
#pragma unroll
    for (int j = 0; j < 16; ++j) { wk[j] = w3[(c0 + j) * kHK + k]; wk2[j] = w3[(c0 + j) * kHK + k2]; }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      acc += (double)q[c0 + j] * (double)wk[j] * (double)wk2[j];
      if (k2 == 0) accv += (double)p[c0 + j] * (double)wk[j];  // v[k] = sum_c p[c] W3[c][k]
    }
  }
  if (half == 1) part[k2] = acc;
  __shared__ double partv;
  if (half == 1 && k2 == 0) partv = accv;
  __syncthreads();
  if (half == 0) {
    m3[k * kHK + k2] = (float)(acc + part[k2]);
    if (k2 == 0) v[k] = (float)(accv + partv);
  }
}

struct Gram256Args {
  int r, total_chunks, chunks_per_cloud, groups;
  const float *y2;                       // (b, 128, r) raw output of the layer below
  const float *sc2, *sh2, *mean2, *invstd2;
  const float *m3, *v;                   // (128, 128), (128)
  const float *w3;                       // (256, 128)
  const float *coef3, *sc3, *sh3;        // (256, 3): a = coef3[3 c]; (256) each
  const uint2 *recs;                     // (b, groups, 256) records of pool_gram256_pack_kernel
  float *dq;                             // (b, 128, r)
  float *stats_part;                     // (128, workgroups, 2)
  float *part_c2, *part_s2, *part_r;     // per workgroup: 16384, 128, 32768 floats
};

// three bf16 terms of an fp32 value (truncation split, exact), as the upper halves of three words
__device__ __forceinline__ void split_terms(float x, unsigned &h, unsigned &m, unsigned &l) {
  const float hf = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
  const float r1 = x - hf;
  const float mf = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
  const float lf = r1 - mf;
  h = __builtin_bit_cast(unsigned, hf) >> 16;
  m = __builtin_bit_cast(unsigned, mf) >> 16;
  l = __builtin_bit_cast(unsigned, lf) >> 16;
}

// ---- before the passes: the pooled tensors (b, 256, groups) as ONE 8-byte record per (group,
// channel), channel fastest: x = winner's sample | hi << 16, y = mid | lo << 16 (the three bf16 terms
// of a * dpooled); sample 0xffff where the ReLU behind the pool was shut.  Both passes read a chunk's
// records as whole cache lines (the tensors themselves put consecutive channels groups * 4 bytes apart)
__global__ void __launch_bounds__(256)
pool_gram256_pack_kernel(int groups, const int *__restrict__ argmax, const float *__restrict__ dpooled,
                         const float *__restrict__ ymax, const float *__restrict__ coef3,
                         const float *__restrict__ sc3, const float *__restrict__ sh3,
                         uint2 *__restrict__ recs) {
  __shared__ uint2 tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, g0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c0 + ty + 8 * e, g = g0 + tx;
    uint2 rec = make_uint2(0xffffu, 0u);
    if (g < groups) {
      const size_t at = ((size_t)b * kHM + c) * groups + g;
      const int am = argmax[at];
      const float dp = dpooled[at], ym = ymax[at];
      unsigned h, m, l;
      split_terms(coef3[c * 3] * dp, h, m, l);
      const bool open = __fmaf_rn(ym, sc3[c], sh3[c]) > 0.f && (unsigned)am < 0xffffu;
      rec = make_uint2((open ? (unsigned)am : 0xffffu) | (h << 16), m | (l << 16));
    }
    tile[ty + 8 * e][tx] = rec;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int g = g0 + ty + 8 * e, c = c0 + tx;
    if (g < groups) recs[((size_t)b * groups + g) * kHM + c] = tile[tx][ty + 8 * e];
  }
}

// one slice (32 rows x 32 columns, a float4 per lane) of a y2 chunk -> relu(bn(.)) -> images
template <bool RAWCOPY>
__device__ __forceinline__ float stage_a2_slice(char *base, int row, int seg_c, const float4 &x,
                                                const RowCoef &rc) {
  const float xv[4] = {x.x, x.y, x.z, x.w};
  float v[4], h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = transform<OP_BNRELU>(xv[e], 0.f, rc);
    h[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[e]) & 0xffff0000u);
    const float r1 = v[e] - h[e];
    m[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
    l[e] = r1 - m[e];
  }
  char *dst = base + (size_t)row * kHRP + seg_c * 2;
  *reinterpret_cast<uint2 *>(dst) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
  *reinterpret_cast<uint2 *>(dst + kHQIMG) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
  *reinterpret_cast<uint2 *>(dst + 2 * kHQIMG) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
  if (RAWCOPY) *reinterpret_cast<float4 *>(base + 3 * kHQIMG + (size_t)row * kHRAWP + seg_c * 4) = x;
  return (v[0] + v[1]) + (v[2] + v[3]);
}

// Scheduling barriers of the data-gradient pass.  G256_SB: around every pair of MFMA steps (pins the
// LDS prefetch of the next pair ahead of this pair's MFMAs); G256_SBS: behind every two MFMAs of the
// sparse part and the staging piece that follows them.  The dense part has the pair-level barriers
// only: with barriers behind each of ITS two-MFMA groups as well, this compiler (ROCm 7.2) produced a
// kernel that returns garbage -- each of the two sets alone is correct and as fast
// (tests/test_gpu_mlp.py::test_pooled_backward_from_the_gram_matrix guards the shipped form).
#define G256_SB() __builtin_amdgcn_sched_barrier(0)
#define G256_SBS() __builtin_amdgcn_sched_barrier(0)
#define G256_SBD() do {} while (0)
// ---- kernel 1: da2 = M3 a2 + v + W3^T S, and the BatchNorm-backward sums of the layer below
template <int NS>
__global__ void __launch_bounds__(256, 1) pool_gram256_dgrad_kernel(const Gram256Args a) {
  constexpr int M = kHM, K = kHK, TN = kHTN, G = TN / NS;
  constexpr int RP = kHRP, QIMG = kHQIMG, RAWP = kHRAWP, SP = kHSP;
  constexpr int SIMG = kHSIMG + kHSP;  // 32 rows + one that nobody reads: where absent entries are written
  constexpr int BUFQ = 3 * QIMG + kHRAW;
  constexpr int STOFF = 2 * BUFQ, RCOFF = STOFF + 3 * SIMG, VOFF = RCOFF + K * 16;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int seg_row = tid >> 3, seg_c = (tid & 7) * 4;
  char *const ST = lds + STOFF;

  for (int t = tid; t < 3 * SIMG / 16; t += 256) reinterpret_cast<uint4 *>(ST)[t] = make_uint4(0u, 0u, 0u, 0u);
  // per row of a2: (sc, sh, mu, is) of the layer below, and v -- in LDS, not in registers (the split
  // fragments of M3 and W3^T take 288 of them)
  if (tid < K) {
    reinterpret_cast<float4 *>(lds + RCOFF)[tid] = make_float4(a.sc2[tid], a.sh2[tid], a.mean2[tid], a.invstd2[tid]);
    reinterpret_cast<float *>(lds + VOFF)[tid] = a.v[tid];
  }
  const unsigned q_lane0 = (unsigned)seg_row * (unsigned)a.r + (unsigned)seg_c;  // + 32 it r per slice
  // this thread's channel of the sparse gradient (256 threads = 256 channels)
  int s_written[G];
#pragma unroll
  for (int gi = 0; gi < G; ++gi) s_written[gi] = 32;

  // A fragments of this wave's 32 rows of da2, split once: M3 (step s: k' = 16 s + 8 lhi + 0..7) and
  // W3^T (step s: channels 16 s + 8 lhi + 0..7); v
  Split3 msp[K / 16], wsp[M / 16];
  {
    const float *mr = a.m3 + (size_t)(32 * wave + l31) * K;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      msp[s] = split3(*reinterpret_cast<const float4 *>(mr + 16 * s + 8 * lhi),
                      *reinterpret_cast<const float4 *>(mr + 16 * s + 8 * lhi + 4));
      // (four steps' loads in flight at a time: with all 24 steps' loads hoisted to the top the
      // kernel's setup spilled ~150 registers AND one build returned garbage rows -- see G256_SB)
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    const float *wc = a.w3 + 32 * wave + l31;
#pragma unroll
    for (int s = 0; s < M / 16; ++s) {
      float w8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w8[j] = wc[(size_t)(16 * s + 8 * lhi + j) * K];
      wsp[s] = split3(w8);
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float st1[16], st2[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) { st1[q] = 0.f; st2[q] = 0.f; }

  const int per = (a.total_chunks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = (int)blockIdx.x * per;
  const int c_hi = c_lo + per < a.total_chunks ? c_lo + per : a.total_chunks;

  float4 qx[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) qx[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  uint2 l_rec[G];
#pragma unroll
  for (int gi = 0; gi < G; ++gi) l_rec[gi] = make_uint2(0xffffu, 0u);

  auto fetch_y = [&](int it, int c) {
    const int b = c / a.chunks_per_cloud;
    const int col0 = (c - b * a.chunks_per_cloud) * TN;
    qx[it] = *reinterpret_cast<const float4 *>(a.y2 + ((size_t)b * K + 32 * it) * a.r + col0 + q_lane0);
  };
  auto fetch_p = [&](int c) {
    const int b = c / a.chunks_per_cloud;
    const int g0 = (c - b * a.chunks_per_cloud) * G;
#pragma unroll
    for (int gi = 0; gi < G; ++gi) l_rec[gi] = a.recs[((size_t)b * a.groups + g0 + gi) * M + tid];
  };
  // (the row's scale / shift come from the LDS table, read by the caller AHEAD of the MFMAs the
  // staging is to hide under)
  auto row_coef = [&](int it) {
    return *reinterpret_cast<const float2 *>(lds + RCOFF + (size_t)(seg_row + 32 * it) * 16);
  };
  auto stage_y = [&](int it, int buf, const float2 c2) {
    const RowCoef rc = {c2.x, c2.y, 0.f, 0.f, 0.f};
    stage_a2_slice<true>(lds + (size_t)buf * BUFQ, seg_row + 32 * it, seg_c, qx[it], rc);
  };
  // the same in five pieces of a few vector instructions (state in sv / sh / sm / sl), one per pair of
  // MFMAs of the sparse part: transform; first term; second and third; the images; the raw copy
  float sv[4], sh[4], sm[4], sl[4];
  auto stage_y_piece = [&](int it, int ph, int buf, const float2 c2) {
    char *base = lds + (size_t)buf * BUFQ;
    const int row = seg_row + 32 * it;
    if (ph == 0) {
      const float xv[4] = {qx[it].x, qx[it].y, qx[it].z, qx[it].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) sv[e] = fmaxf(__fmaf_rn(xv[e], c2.x, c2.y), 0.f);
    } else if (ph == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sh[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sv[e]) & 0xffff0000u);
        sv[e] = sv[e] - sh[e];
      }
    } else if (ph == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sm[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sv[e]) & 0xffff0000u);
        sl[e] = sv[e] - sm[e];
      }
    } else if (ph == 3) {
      char *dst = base + (size_t)row * RP + seg_c * 2;
      *reinterpret_cast<uint2 *>(dst) = make_uint2(pack_hi16(sh[0], sh[1]), pack_hi16(sh[2], sh[3]));
      *reinterpret_cast<uint2 *>(dst + QIMG) = make_uint2(pack_hi16(sm[0], sm[1]), pack_hi16(sm[2], sm[3]));
      *reinterpret_cast<uint2 *>(dst + 2 * QIMG) = make_uint2(pack_hi16(sl[0], sl[1]), pack_hi16(sl[2], sl[3]));
    } else {
      *reinterpret_cast<float4 *>(base + 3 * QIMG + (size_t)row * RAWP + seg_c * 4) = qx[it];
    }
  };
  // the S^T image: this thread's entries of the chunk just consumed are cleared, the next chunk's
  // written (real: that chunk exists); an entry that does not exist goes to row 32, which nobody reads
  // -- no branches, so that the pieces can sit between the dense part's MFMAs
  auto clear_s = [&](int gi) {
    unsigned short *dst = reinterpret_cast<unsigned short *>(ST + (size_t)s_written[gi] * SP) + tid;
    dst[0] = 0; dst[SIMG / 2] = 0; dst[SIMG] = 0;
  };
  auto write_s = [&](int gi, bool real) {
    const unsigned am = l_rec[gi].x & 0xffffu;
    const int n = (real && am < (unsigned)NS) ? gi * NS + (int)am : 32;
    s_written[gi] = n;
    unsigned short *dst = reinterpret_cast<unsigned short *>(ST + (size_t)n * SP) + tid;
    dst[0] = (unsigned short)(l_rec[gi].x >> 16);
    dst[SIMG / 2] = (unsigned short)(l_rec[gi].y & 0xffffu);
    dst[SIMG] = (unsigned short)(l_rec[gi].y >> 16);
  };
  auto clampc = [&](int c) { return c < c_hi ? c : c_hi - 1; };
  __syncthreads();  // the zero fill is complete
  if (c_lo < c_hi) {
#pragma unroll
    for (int it = 0; it < 4; ++it) fetch_y(it, c_lo);
    fetch_p(c_lo);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      stage_y(it, 0, row_coef(it));
      fetch_y(it, clampc(c_lo + 1));
    }
#pragma unroll
    for (int gi = 0; gi < G; ++gi) write_s(gi, true);
    fetch_p(clampc(c_lo + 1));
  }
  __syncthreads();

  const int tr_off = (8 * lhi + ((lane & 15) >> 2)) * RP + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int st_off = l31 * SP + 8 * lhi * 2;

  auto chunk = [&](auto curt, int c) {
    constexpr int cur = decltype(curt)::value;
    const char *Qc = lds + (size_t)cur * BUFQ;
    const int ahead = clampc(c + 2);
    const int b = c / a.chunks_per_cloud;
    const int col0 = (c - b * a.chunks_per_cloud) * TN;
    f32x16 accD, accE;
#pragma unroll
    for (int q = 0; q < 16; ++q) { accD[q] = 0.f; accE[q] = 0.f; }
#define GD_STEP(A0, A1, B0, B1, AT, BT)                                                              \
  accD = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A0).AT, (B0).BT, accD, 0, 0, 0);                   \
  accE = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A1).AT, (B1).BT, accE, 0, 0, 0)
#define GD_PAIR(A0, A1, B0, B1)                                                                      \
  GD_STEP(A0, A1, B0, B1, lo, hi); GD_STEP(A0, A1, B0, B1, hi, lo); GD_STEP(A0, A1, B0, B1, mid, mid); \
  GD_STEP(A0, A1, B0, B1, mid, hi); GD_STEP(A0, A1, B0, B1, hi, mid); GD_STEP(A0, A1, B0, B1, hi, hi)
    // ---- sparse part: W3^T (registers) * S (row-wise 16-byte reads of S^T), two steps at a time
    auto sfrag = [&](int s) {
      Split3 f;
      const char *p0 = ST + st_off + 16 * s * 2;
      f.hi = *reinterpret_cast<const bf16x8 *>(p0);
      f.mid = *reinterpret_cast<const bf16x8 *>(p0 + SIMG);
      f.lo = *reinterpret_cast<const bf16x8 *>(p0 + 2 * SIMG);
      return f;
    };
    {
      // (the scheduling barriers pin the fragment reads of the NEXT pair of steps ahead of this pair's
      // MFMAs: at one wave per SIMD nothing else hides an LDS round trip, and left to itself the
      // scheduler issues every read right in front of its MFMA)
      Split3 b0 = sfrag(0), b1 = sfrag(1);
#pragma unroll
      for (int s = 0; s < M / 16; s += 2) {
        const Split3 c0 = b0, c1 = b1;
        if (s + 2 < M / 16) { b0 = sfrag(s + 2); b1 = sfrag(s + 3); }
        const float2 rc2 = row_coef(s / 2 < 4 ? s / 2 : 0);
        G256_SB();
        // The pair's twelve MFMAs two at a time, a piece of the next chunk's staging behind each two
        // and a scheduling barrier behind that: left alone the scheduler issues the MFMAs back to back
        // and the staging after them, where nothing is in flight to hide it (one wave per SIMD)
        const bool stg = s / 2 < 4;
        const int it = s / 2 < 4 ? s / 2 : 0;
        GD_STEP(wsp[s], wsp[s + 1], c0, c1, lo, hi);
        if (stg) stage_y_piece(it, 0, cur ^ 1, rc2);
        G256_SBS();
        GD_STEP(wsp[s], wsp[s + 1], c0, c1, hi, lo);
        if (stg) stage_y_piece(it, 1, cur ^ 1, rc2);
        G256_SBS();
        GD_STEP(wsp[s], wsp[s + 1], c0, c1, mid, mid);
        if (stg) stage_y_piece(it, 2, cur ^ 1, rc2);
        G256_SBS();
        GD_STEP(wsp[s], wsp[s + 1], c0, c1, mid, hi);
        if (stg) stage_y_piece(it, 3, cur ^ 1, rc2);
        G256_SBS();
        GD_STEP(wsp[s], wsp[s + 1], c0, c1, hi, mid);
        if (stg) stage_y_piece(it, 4, cur ^ 1, rc2);
        G256_SBS();
        GD_STEP(wsp[s], wsp[s + 1], c0, c1, hi, hi);
        if (stg) fetch_y(it, ahead);
        G256_SB();
      }
    }
    // the first fragments of the dense part (the a2 images of this chunk: nobody writes them now)
    h_bf16x4 pf[2][2][3][2];
    auto frag = [&](int s, h_bf16x4 (&dst)[3][2]) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const char *p0 = Qc + (size_t)t * QIMG + (size_t)(16 * s) * RP + tr_off;
        dst[t][0] = h_lds_read_tr(p0);
        dst[t][1] = h_lds_read_tr(p0 + 4 * RP);
      }
    };
    frag(0, pf[0][0]);
    frag(1, pf[0][1]);
    G256_SB();
    __syncthreads();  // S^T of chunk c read by every wave
    // ---- dense part: M3 (registers) * a2 (transposing reads of the images); behind its first MFMAs
    // the entries of S^T are cleared, the next chunk's written, the one after's requested
    {
      auto operand = [&](const h_bf16x4 (&src)[3][2]) {
        Split3 sb;
        sb.hi = __builtin_shufflevector(src[0][0], src[0][1], 0, 1, 2, 3, 4, 5, 6, 7);
        sb.mid = __builtin_shufflevector(src[1][0], src[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
        sb.lo = __builtin_shufflevector(src[2][0], src[2][1], 0, 1, 2, 3, 4, 5, 6, 7);
        return sb;
      };
      const bool next_real = c + 1 < c_hi;
#pragma unroll
      for (int s = 0; s < K / 16; s += 2) {
        const int ring = (s >> 1) & 1;
        if (s + 2 < K / 16) {
          frag(s + 2, pf[ring ^ 1][0]);
          frag(s + 3, pf[ring ^ 1][1]);
        }
        G256_SB();
        const Split3 c0 = operand(pf[ring][0]), c1 = operand(pf[ring][1]);
        GD_STEP(msp[s], msp[s + 1], c0, c1, lo, hi);
        if (s == 0) clear_s(0);
        if (s == 2 && G > 1) clear_s(G - 1);
        G256_SBD();
        GD_STEP(msp[s], msp[s + 1], c0, c1, hi, lo);
        G256_SBD();
        GD_STEP(msp[s], msp[s + 1], c0, c1, mid, mid);
        if (s == 0) write_s(0, next_real);
        if (s == 2 && G > 1) write_s(G - 1, next_real);
        G256_SBD();
        GD_STEP(msp[s], msp[s + 1], c0, c1, mid, hi);
        G256_SBD();
        GD_STEP(msp[s], msp[s + 1], c0, c1, hi, mid);
        if (s == 4) fetch_p(ahead);
        G256_SBD();
        GD_STEP(msp[s], msp[s + 1], c0, c1, hi, hi);
        G256_SB();
      }
    }
#undef GD_PAIR
#undef GD_STEP
    // ---- the tile leaves; the BatchNorm-backward sums of the layer below from it
    {
      float *dst = a.dq + ((size_t)b * K + 32 * wave + 4 * lhi) * a.r + col0 + l31;
      const float4 *rc = reinterpret_cast<const float4 *>(lds + RCOFF);
      const char *raw = Qc + 3 * QIMG;
#pragma unroll
      for (int q4 = 0; q4 < 16; q4 += 4) {  // four rows at a time: their LDS reads first, in one batch
        float4 c4[4];
        float yv[4], vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 32 * wave + 4 * lhi + e + 8 * (q4 >> 2);
          c4[e] = rc[row];
          yv[e] = *reinterpret_cast<const float *>(raw + (size_t)row * RAWP + l31 * 4);
          vv[e] = reinterpret_cast<const float *>(lds + VOFF)[row];
        }
        G256_SB();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = q4 + e, ro = e + 8 * (q4 >> 2);
          const float d = (accD[q] + accE[q]) + vv[e];
          __builtin_nontemporal_store(d, &dst[(size_t)ro * a.r]);
          const float gg = __fmaf_rn(yv[e], c4[e].x, c4[e].y) > 0.f ? d : 0.f;
          st1[q] += gg;
          st2[q] = __fmaf_rn(gg, (yv[e] - c4[e].z) * c4[e].w, st2[q]);
        }
      }
    }
    __syncthreads();  // chunk c read by everyone, chunk c+1 staged by everyone
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  for (int c = c_lo; c < c_hi; c += 2) {
    chunk(B0{}, c);
    if (c + 1 < c_hi) chunk(B1{}, c + 1);
  }

  // the lanes' column sums -> row sums through LDS (each wave parks and reads only its own rows)
  {
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
}

// ---- kernel 2: R = S a2^T, the upper blocks of C2 = a2 a2^T, s2 = sum a2
template <int NS>
__global__ void __launch_bounds__(256, 1) pool_gram256_wgrad_kernel(const Gram256Args a) {
  constexpr int M = kHM, K = kHK, TN = kHTN, G = TN / NS;
  constexpr int RP = kHRP, QIMG = kHQIMG;
  constexpr int BUFQ = 3 * QIMG;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int seg_row = tid >> 3, seg_c = (tid & 7) * 4;

  RowCoef qc[4];
  size_t q_lane[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    qc[it] = {a.sc2[seg_row + 32 * it], a.sh2[seg_row + 32 * it], 0.f, 0.f, 0.f};
    q_lane[it] = (size_t)(seg_row + 32 * it) * a.r + seg_c;
  }
  float s2acc[4] = {0.f, 0.f, 0.f, 0.f};
  f32x16 accR[2][4], accC[3];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) accR[i][j][q] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) accC[j][q] = 0.f;
  }

  const int per = (a.total_chunks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = (int)blockIdx.x * per;
  const int c_hi = c_lo + per < a.total_chunks ? c_lo + per : a.total_chunks;

  float4 qx[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) qx[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  // pooled tensors of the lane's channels: raw values of the NEXT chunk, the current chunk's as the
  // winner's column in the chunk (-1: none) and the three bf16 terms of a * dpooled
  uint2 n_rec[2][G];
  int c_n[2][G];
  unsigned c_h[2][G], c_m[2][G], c_l[2][G];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      n_rec[i][gi] = make_uint2(0xffffu, 0u);
      c_n[i][gi] = -1; c_h[i][gi] = 0u; c_m[i][gi] = 0u; c_l[i][gi] = 0u;
    }

  auto fetch_y = [&](int it, int c) {
    const int b = c / a.chunks_per_cloud;
    const int col0 = (c - b * a.chunks_per_cloud) * TN;
    qx[it] = *reinterpret_cast<const float4 *>(a.y2 + (size_t)b * K * a.r + col0 + q_lane[it]);
  };
  auto fetch_p = [&](int c) {
    const int b = c / a.chunks_per_cloud;
    const int g0 = (c - b * a.chunks_per_cloud) * G;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int gi = 0; gi < G; ++gi)
        n_rec[i][gi] = a.recs[((size_t)b * a.groups + g0 + gi) * M + 64 * wave + 32 * i + l31];
  };
  auto adopt_p = [&](bool real) {  // next -> current
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        const unsigned am = n_rec[i][gi].x & 0xffffu;
        c_n[i][gi] = (real && am < (unsigned)NS) ? gi * NS + (int)am : -1;
        c_h[i][gi] = n_rec[i][gi].x >> 16;
        c_m[i][gi] = n_rec[i][gi].y & 0xffffu;
        c_l[i][gi] = n_rec[i][gi].y >> 16;
      }
  };
  auto stage_y = [&](int it, int buf, bool real) {
    const float s = stage_a2_slice<false>(lds + (size_t)buf * BUFQ, seg_row + 32 * it, seg_c, qx[it], qc[it]);
    if (real) s2acc[it] += s;
  };
  auto clampc = [&](int c) { return c < c_hi ? c : c_hi - 1; };
  if (c_lo < c_hi) {
#pragma unroll
    for (int it = 0; it < 4; ++it) fetch_y(it, c_lo);
    fetch_p(c_lo);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      stage_y(it, 0, true);
      fetch_y(it, clampc(c_lo + 1));
    }
    adopt_p(true);
    fetch_p(clampc(c_lo + 1));
  }
  __syncthreads();

  const int rw_off = l31 * RP + 8 * lhi * 2;

  auto chunk = [&](auto curt, int c) {
    constexpr int cur = decltype(curt)::value;
    const char *Qc = lds + (size_t)cur * BUFQ;
    const int ahead = clampc(c + 2);
    auto rows = [&](int blk, int s) {
      Split3 f;
      const char *p0 = Qc + (size_t)(blk * 32) * RP + rw_off + 16 * s * 2;
      f.hi = *reinterpret_cast<const bf16x8 *>(p0);
      f.mid = *reinterpret_cast<const bf16x8 *>(p0 + QIMG);
      f.lo = *reinterpret_cast<const bf16x8 *>(p0 + 2 * QIMG);
      return f;
    };
    // the lane's fragment of S for row block i, step s: its channel's entry if the winner's column
    // lies among the lane's eight columns 16 s + 8 lhi + 0..7
    auto onehot = [&](int i, int s) {
      const int gi = G == 1 ? 0 : s;  // NS = 16: step s covers group s of the chunk
      const int p = c_n[i][gi] - (16 * s + 8 * lhi);
      const bool in = (unsigned)p < 8u;
      const int d = p >> 1, sh = (p & 1) * 16;
      const unsigned vh = c_h[i][gi] << sh, vm = c_m[i][gi] << sh, vl = c_l[i][gi] << sh;
      h_u32x4 uh, um, ul;
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        const bool at = in && d == dd;
        uh[dd] = at ? vh : 0u; um[dd] = at ? vm : 0u; ul[dd] = at ? vl : 0u;
      }
      Split3 f;
      f.hi = __builtin_bit_cast(bf16x8, uh);
      f.mid = __builtin_bit_cast(bf16x8, um);
      f.lo = __builtin_bit_cast(bf16x8, ul);
      return f;
    };
#pragma unroll
    for (int s = 0; s < TN / 16; ++s) {
      Split3 sq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) sq[j] = rows(j, s);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const Split3 sp = onehot(i, s);
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma_x6(accR[i][j], sp, sq[j]);
        // the next chunk's a2: staged into the other buffer, the one after requested
        stage_y(2 * s + i, cur ^ 1, c + 1 < c_hi);
        fetch_y(2 * s + i, ahead);
      }
      // upper blocks of the Gram matrix: wave 0: (0,0) (0,1) (0,2); 1: (1,1) (1,2) (1,3);
      // 2: (2,2) (2,3); 3: (3,3) (0,3)
      if (wave == 0) {
        mfma_x6(accC[0], sq[0], sq[0]); mfma_x6(accC[1], sq[0], sq[1]); mfma_x6(accC[2], sq[0], sq[2]);
      } else if (wave == 1) {
        mfma_x6(accC[0], sq[1], sq[1]); mfma_x6(accC[1], sq[1], sq[2]); mfma_x6(accC[2], sq[1], sq[3]);
      } else if (wave == 2) {
        mfma_x6(accC[0], sq[2], sq[2]); mfma_x6(accC[1], sq[2], sq[3]);
      } else {
        mfma_x6(accC[0], sq[3], sq[3]); mfma_x6(accC[1], sq[0], sq[3]);
      }
    }
    adopt_p(c + 1 < c_hi);
    fetch_p(ahead);
    __syncthreads();  // chunk c read by everyone, chunk c+1 staged by everyone
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  for (int c = c_lo; c < c_hi; c += 2) {
    chunk(B0{}, c);
    if (c + 1 < c_hi) chunk(B1{}, c + 1);
  }

  // ---- per-workgroup partials
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    float s = s2acc[it];
    s += __shfl_xor(s, 1, kWave);
    s += __shfl_xor(s, 2, kWave);
    s += __shfl_xor(s, 4, kWave);
    if ((tid & 7) == 0) a.part_s2[(size_t)blockIdx.x * K + seg_row + 32 * it] = s;
  }
  float *orr = a.part_r + (size_t)blockIdx.x * M * K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ro = (q & 3) + 8 * (q >> 2) + 4 * lhi;
        orr[(size_t)(64 * wave + 32 * i + ro) * K + 32 * j + l31] = accR[i][j][q];
      }
  float *oc = a.part_c2 + (size_t)blockIdx.x * K * K;
  auto put = [&](const f32x16 &acc, int bi, int bj) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int ro = (q & 3) + 8 * (q >> 2) + 4 * lhi;
      oc[(size_t)(32 * bi + ro) * K + 32 * bj + l31] = acc[q];
    }
  };
  if (wave == 0) { put(accC[0], 0, 0); put(accC[1], 0, 1); put(accC[2], 0, 2); }
  else if (wave == 1) { put(accC[0], 1, 1); put(accC[1], 1, 2); put(accC[2], 1, 3); }
  else if (wave == 2) { put(accC[0], 2, 2); put(accC[1], 2, 3); }
  else { put(accC[0], 3, 3); put(accC[1], 0, 3); }
}

// C2 (128 x 128, upper blocks), s2 (128) and R (256 x 128) from the workgroups' partials, in double
constexpr int kHSums = kHK * kHK + kHK + kHM * kHK;  // 49 280 = 1540 x 32

__global__ void __launch_bounds__(256)
pool_gram256_reduce_kernel(int parts, const float *__restrict__ part_c2, const float *__restrict__ part_s2,
                           const float *__restrict__ part_r, double *__restrict__ sums) {
  __shared__ double red[8][32];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
  const float *src;
  size_t stride;
  bool live = true;
  if (e < kHK * kHK) {
    src = part_c2 + e; stride = kHK * kHK;
    live = (e / kHK) / 32 <= (e % kHK) / 32;  // (the lower blocks are never written)
  } else if (e < kHK * kHK + kHK) { src = part_s2 + (e - kHK * kHK); stride = kHK; }
  else { src = part_r + (e - kHK * kHK - kHK); stride = kHM * kHK; }
  double s = 0.0;
  if (live) {
    int p = sl;
    for (; p + 24 < parts; p += 32) {
      const float v0 = src[(size_t)p * stride], v1 = src[(size_t)(p + 8) * stride];
      const float v2 = src[(size_t)(p + 16) * stride], v3 = src[(size_t)(p + 24) * stride];
      s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; p < parts; p += 8) s += (double)src[(size_t)p * stride];
  }
  red[sl][threadIdx.x & 31] = s;
  __syncthreads();
  if (sl == 0) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][threadIdx.x];
    sums[e] = t;
  }
}

// dW3[c][k] = q_c sum_k' W3[c][k'] C2[k'][k] + p_c s2[k] + R[c][k]   (C2 symmetric: upper blocks);
// one workgroup per channel c, thread (k = tid & 127, half = tid >> 7) sums half of the k'
__global__ void __launch_bounds__(256)
pool_gram256_dw_kernel(const float *__restrict__ w3, const float *__restrict__ qp,
                       const double *__restrict__ sums, float *__restrict__ dw) {
  __shared__ double part[kHK];
  const int c = blockIdx.x, k = threadIdx.x & (kHK - 1), half = threadIdx.x >> 7;
  const double *c2 = sums, *s2 = sums + kHK * kHK, *rr = s2 + kHK;
  double acc = 0.0;
  for (int k0 = half * (kHK / 2); k0 < (half + 1) * (kHK / 2); k0 += 16) {
    double g[16];
    float w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k2 = k0 + j;
      g[j] = (k2 >> 5) <= (k >> 5) ? c2[k2 * kHK + k] : c2[k * kHK + k2];
      w[j] = w3[c * kHK + k2];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += (double)w[j] * g[j];
  }
  if (half == 1) part[k] = acc;
  __syncthreads();
  if (half == 0)
    dw[c * kHK + k] = (float)((double)qp[c * 2] * (acc + part[k]) + (double)qp[c * 2 + 1] * s2[k] + rr[c * kHK + k]);
}

int gram256_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}

// workgroups of the two passes: one per CU, at least `least` chunks each (the weight pass writes
// 197 KB of partials per workgroup: it takes fewer, longer ranges)
int gram256_workgroups(int b, int r, int least) {
  const long long total = (long long)b * (r / 32);
  long long g = gram256_cus();
  if (g > total / least) g = total / least;
  return (int)(g < 1 ? 1 : g);
}
int gram256_dgrad_least() {
  static const int v = getenv("MLP_GRAM256_DGRAD_CHUNKS") ? atoi(getenv("MLP_GRAM256_DGRAD_CHUNKS")) : 8;
  return v > 0 ? v : 8;
}
int gram256_wgrad_least() {
  static const int v = getenv("MLP_GRAM256_WGRAD_CHUNKS") ? atoi(getenv("MLP_GRAM256_WGRAD_CHUNKS")) : 16;
  return v > 0 ? v : 16;
}

constexpr size_t kDgradLds = 2 * (3 * (size_t)kHQIMG + kHRAW) + 3 * (size_t)(kHSIMG + kHSP) + kHK * 16 + kHK * 4;
constexpr size_t kWgradLds = 2 * 3 * (size_t)kHQIMG;
static_assert(kDgradLds <= 160 * 1024, "LDS of the data-gradient pass");

}  // namespace

#define MLP_API extern "C" __attribute__((visibility("default")))

// 1 when mlp_pool_gram256_backward covers the layer: (m, k) = (256, 128), nsample 16 / 32, whole
// 32-column chunks per cloud
MLP_API int mlp_pool_gram256_supported(int b, int m, int k, int r, int ns) {
  static const bool off = (getenv("MLP_POOL_GRAM") && atoi(getenv("MLP_POOL_GRAM")) == 0) ||
                          (getenv("MLP_POOL_GRAM256") && atoi(getenv("MLP_POOL_GRAM256")) == 0);
  // worth it from SA2's size on (B = 8: 8192 chunks; measured: 308 us against 355 there, 154 against
  // 105 at SA3's 2048 chunks -- the two passes' fixed costs); read on every call: tests lower it
  const long long least = getenv("MLP_POOL_GRAM256_MIN_CHUNKS") ? atoll(getenv("MLP_POOL_GRAM256_MIN_CHUNKS")) : 4096;
  if (off || b <= 0 || m != kHM || k != kHK || r <= 0 || r % 32 != 0) return 0;
  if ((ns != 16 && ns != 32) || r % ns != 0) return 0;
  return (long long)b * (r / 32) >= least ? 1 : 0;
}

// number of per-workgroup partials of stats_part (128, parts, 2) and floats of workspace
MLP_API int mlp_pool_gram256_parts(int b, int r) { return gram256_workgroups(b, r, gram256_dgrad_least()); }
MLP_API size_t mlp_pool_gram256_workspace_floats(int b, int r, int ns) {
  const size_t g = (size_t)gram256_workgroups(b, r, gram256_wgrad_least());
  const size_t groups = ns > 0 ? (size_t)(r / ns) : 0;
  // qp (512) + M3 (16384) + v (128) + the records (2 words per group and channel) + partials per
  // workgroup + the sums as doubles
  return 512 + 16384 + 128 + (size_t)b * groups * kHM * 2 + g * (size_t)kHSums + 2 * (size_t)kHSums + 16;
}

// Backward of the pooled last layer y3 = w3 . relu(bn2(y2)), w3 (256,128), from y2 and the pooled
// tensors alone: dq (b,128,r) = gradient w.r.t. relu(bn2(y2)); dw3 (256,128); stats_part
// (128, parts, 2): the BatchNorm-backward sums of layer 2 (for mlp_bn_backward_finalize).  coef3
// (256,3) = (a, c1, c2) of layer 3 as mlp_bn_relu_pool_backward leaves them.
MLP_API int mlp_pool_gram256_backward(int b, int r, int ns, const float *w3, const float *y2, const float *sc2,
                                      const float *sh2, const float *mean2, const float *invstd2,
                                      const float *coef3, const float *sc3, const float *sh3,
                                      const float *mean3, const float *invstd3, const int *argmax,
                                      const float *dpooled, const float *ymax, float *dq, float *dw3,
                                      float *stats_part, float *workspace, void *stream_) {
  if (!mlp_pool_gram256_supported(b, kHM, kHK, r, ns) || !w3 || !y2 || !sc2 || !sh2 || !mean2 || !invstd2 ||
      !coef3 || !sc3 || !sh3 || !mean3 || !invstd3 || !argmax || !dpooled || !ymax || !dq || !dw3 ||
      !stats_part || !workspace || (reinterpret_cast<size_t>(workspace) & 15) ||
      (reinterpret_cast<size_t>(y2) & 15))
    return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  const int g1 = gram256_workgroups(b, r, gram256_dgrad_least());
  const int g2 = gram256_workgroups(b, r, gram256_wgrad_least());
  float *qp = workspace, *m3 = qp + 512, *v = m3 + 16384;
  const int groups = r / ns;
  uint2 *recs = reinterpret_cast<uint2 *>(v + 128);  // (16-byte aligned: 17 024 floats in)
  float *part_c2 = v + 128 + (size_t)b * groups * kHM * 2, *part_s2 = part_c2 + (size_t)g2 * kHK * kHK, *part_r = part_s2 + (size_t)g2 * kHK;
  float *tail = part_r + (size_t)g2 * kHM * kHK;
  double *sums = reinterpret_cast<double *>(tail + ((reinterpret_cast<size_t>(tail) & 7) ? 1 : 0));
  hipLaunchKernelGGL(pool_gram256_prep_kernel, dim3(kHK), dim3(256), 0, stream, w3, coef3, mean3,
                     invstd3, qp, m3, v);
  hipLaunchKernelGGL(pool_gram256_pack_kernel, dim3((groups + 31) / 32, kHM / 32, b), dim3(256), 0, stream, groups,
                     argmax, dpooled, ymax, coef3, sc3, sh3, recs);
  static std::mutex mu;
  static bool attr_set = false;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pool_gram256_dgrad_kernel<16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDgradLds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pool_gram256_dgrad_kernel<32>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDgradLds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pool_gram256_wgrad_kernel<16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgradLds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pool_gram256_wgrad_kernel<32>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgradLds);
      attr_set = true;
    }
  }
  Gram256Args a = {};
  a.r = r; a.total_chunks = b * (r / 32); a.chunks_per_cloud = r / 32; a.groups = groups;
  a.y2 = y2; a.sc2 = sc2; a.sh2 = sh2; a.mean2 = mean2; a.invstd2 = invstd2;
  a.m3 = m3; a.v = v; a.w3 = w3; a.coef3 = coef3; a.sc3 = sc3; a.sh3 = sh3;
  a.recs = recs;
  a.dq = dq; a.stats_part = stats_part; a.part_c2 = part_c2; a.part_s2 = part_s2; a.part_r = part_r;
  if (ns == 16) {
    hipLaunchKernelGGL(pool_gram256_dgrad_kernel<16>, dim3(g1), dim3(256), kDgradLds, stream, a);
    hipLaunchKernelGGL(pool_gram256_wgrad_kernel<16>, dim3(g2), dim3(256), kWgradLds, stream, a);
  } else {
    hipLaunchKernelGGL(pool_gram256_dgrad_kernel<32>, dim3(g1), dim3(256), kDgradLds, stream, a);
    hipLaunchKernelGGL(pool_gram256_wgrad_kernel<32>, dim3(g2), dim3(256), kWgradLds, stream, a);
  }
  static_assert(kHSums % 32 == 0, "whole reduce workgroups");
  hipLaunchKernelGGL(pool_gram256_reduce_kernel, dim3(kHSums / 32), dim3(256), 0, stream, g2, part_c2, part_s2,
                     part_r, sums);
  hipLaunchKernelGGL(pool_gram256_dw_kernel, dim3(kHM), dim3(256), 0, stream, w3, qp, sums, dw3);
  return pn2_launch_status();
}
