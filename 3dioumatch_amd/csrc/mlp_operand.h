// 3dioumatch_amd/csrc/mlp_operand.h -- what the operands of the shared-MLP GEMM kernels ARE:
// tensors transformed on their way from HBM into LDS (the BatchNorm / ReLU algebra of
// pointnet2/pytorch_utils.py:70-124 and its backward), shared by mlp_gemm.hip and
// mlp_bwd_fused.hip.
#pragma once
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));


enum OperandMode { OP_DIRECT = 0, OP_BNRELU = 1, OP_DY = 2, OP_POOLDY = 3, OP_LIN4 = 4 };

struct OperandB {
  const float *x;        // OP_DIRECT / OP_BNRELU: the tensor; OP_DY: y
  const float *dz;       // OP_DY only
  const float *scale;    // per row k
  const float *shift;
  const float *mean;     // OP_DY
  const float *invstd;   // OP_DY
  const float *coef;     // OP_DY: [k][3] = a, c1, c2
  const int *argmax;     // OP_POOLDY: (rows, r/ns) winning sample per group; dz holds dpooled
  int ns;                // OP_POOLDY: samples per group
  int groups;            // OP_POOLDY: r / ns (groups per row)
  const float *lin_w;    // OP_LIN4: (rows, 4) weight of the 4 -> rows layer whose output the
                         // operand stands for; x is then that layer's 4-channel input (b,4,r)
};

constexpr bool is_dy(int mode) { return mode == OP_DY || mode == OP_POOLDY; }

// OP_LIN4: the output of a 4 -> rows first layer is never stored; whoever needs row k at column
// n recomputes y[k][n] = W[k] . x[:, n] with THIS function (one fixed order, so that the ReLU
// gate is the same bit pattern in the forward and in every backward kernel)
__device__ __forceinline__ float lin4(const float4 w, float x0, float x1, float x2, float x3) {
  return __fmaf_rn(w.w, x3, __fmaf_rn(w.z, x2, __fmaf_rn(w.y, x1, w.x * x0)));
}

// Per-row constants of an operand (loaded once per row, kept in registers).  The BatchNorm+ReLU
// backward  a*(g - c1 - ((x - mu)*is)*c2),  g = [x*sc + sh > 0] ? dz : 0,  is affine in x next
// to the gated term:  a*g + (q*x + p)  with  q = -a*is*c2,  p = a*(is*c2*mu - c1)  -- folded per
// row here, so that an element costs four fused multiply-adds / selects instead of ten
// operations (these kernels spend a large share of their issue slots on operand transforms).
struct RowCoef { float sc, sh, a, q, p; };

template <int MODE>
__device__ __forceinline__ RowCoef load_row_coef(const OperandB &op, int k, bool valid) {
  RowCoef c = {1.f, 0.f, 0.f, 0.f, 0.f};
  if (MODE == OP_DIRECT || !valid) return c;
  c.sc = op.scale[k]; c.sh = op.shift[k];
  if (is_dy(MODE)) {
    const float mu = op.mean[k], is = op.invstd[k];
    const float a = op.coef[k * 3], c1 = op.coef[k * 3 + 1], c2 = op.coef[k * 3 + 2];
    const float t = is * c2;
    c.a = a;
    c.q = -(a * t);
    c.p = a * (t * mu - c1);
  }
  return c;
}

template <int MODE>
__device__ __forceinline__ float transform(float x, float dz, const RowCoef &c) {
  if (MODE == OP_DIRECT) return x;
  const float z = __fmaf_rn(x, c.sc, c.sh);
  if (MODE == OP_BNRELU) return fmaxf(z, 0.f);
  const float lin = __fmaf_rn(c.q, x, c.p);
  return z > 0.f ? __fmaf_rn(c.a, dz, lin) : lin;
}

// raw loads of N consecutive elements (x, and dz for OP_DY); zero outside the row / limit
// (`row` = index of the operand row over the whole batch, b * rows + row: OP_POOLDY only)
template <int MODE, int N>
__device__ __forceinline__ void load_raw_segment(const OperandB &op, size_t off, int gr, int limit,
                                                 bool vec_ok, bool row_ok, float *x, float *dz,
                                                 int row = 0) {
#pragma unroll
  for (int i = 0; i < N; ++i) { x[i] = 0.f; dz[i] = 0.f; }
  if (!row_ok) return;
  if (MODE == OP_POOLDY) {
    // dz of the segment from the pooled tensors (B, rows, groups): no 64-bit division here
    const size_t gbase = (size_t)row * op.groups;
    if (op.ns % N == 0 && gr % N == 0) {  // the whole segment lies in one group
      if (gr < limit) {
        const int g = gr / op.ns, s0 = gr - g * op.ns;
        const int win = op.argmax[gbase + g] - s0;
        const float dp = op.dz[gbase + g];
#pragma unroll
        for (int i = 0; i < N; ++i) dz[i] = (i == win) ? dp : 0.f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (gr + i < limit) {
          const int g = (gr + i) / op.ns;
          dz[i] = (op.argmax[gbase + g] == gr + i - g * op.ns) ? op.dz[gbase + g] : 0.f;
        }
      }
    }
  }
  if (vec_ok) {
#pragma unroll
    for (int i = 0; i < N; i += 4) {
      if (gr + i < limit) {
        const float4 v = *reinterpret_cast<const float4 *>(op.x + off + i);
        x[i] = v.x; x[i + 1] = v.y; x[i + 2] = v.z; x[i + 3] = v.w;
        if (MODE == OP_DY) {
          const float4 d = *reinterpret_cast<const float4 *>(op.dz + off + i);
          dz[i] = d.x; dz[i + 1] = d.y; dz[i + 2] = d.z; dz[i + 3] = d.w;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (gr + i < limit) {
        x[i] = op.x[off + i];
        if (MODE == OP_DY) dz[i] = op.dz[off + i];
      }
    }
  }
}

// ---- fp32 products on the bf16 matrix pipe -----------------------------------------------------
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate.  An fp32 value is the EXACT sum of three
// bf16 values (truncation split: hi = the top 8 significand bits, mid = the next 8, lo = the
// last 8; each difference is exact), a bf16 x bf16 product is exact in fp32, and the matrix pipe
// accumulates in fp32.  Of the nine partial products of a * b the three below 2^-24 |a b|
// (mid*lo, lo*mid, lo*lo) are dropped -- the same order as one fp32 rounding -- and the other six
// are issued as v_mfma_f32_32x32x16_bf16: 6 x 32 cycles per 16 k instead of 8 x 64 (2.7x less
// matrix time) for fp32-grade results (max error over the layer shapes 1e-7 of the output
// range, profiles/r4_split_bf16.json).  A lane's MFMA operand (row / column lane & 31, k half
// lane >> 5, eight consecutive k) is exactly what the k-quad LDS layout hands it in two 16-byte
// reads, so the split happens on the fragments, in registers.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
struct Split3 { bf16x8 hi, mid, lo; };

__device__ __forceinline__ unsigned pack_hi16(float a, float b) {  // (b.hi16 << 16) | a.hi16
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}

__device__ __forceinline__ Split3 split3(const float4 &u, const float4 &v) {
  const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
  float h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[e]) & 0xffff0000u);
    const float r1 = x[e] - h[e];                       // exact
    m[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
    l[e] = r1 - m[e];                                   // exact, at most 8 significant bits
  }
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  u32x4v ph, pm, pl;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ph[e] = pack_hi16(h[2 * e], h[2 * e + 1]);
    pm[e] = pack_hi16(m[2 * e], m[2 * e + 1]);
    pl[e] = pack_hi16(l[2 * e], l[2 * e + 1]);
  }
  Split3 o;
  o.hi = __builtin_bit_cast(bf16x8, ph);
  o.mid = __builtin_bit_cast(bf16x8, pm);
  o.lo = __builtin_bit_cast(bf16x8, pl);
  return o;
}


// the same split of eight values given one by one (fragments gathered with 4-byte LDS reads)
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
  return split3(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]));
}

// acc += a * b over the lane group's 16 k, fp32-grade: the six significant partial products,
// small ones first
__device__ __forceinline__ void mfma_x6(f32x16 &acc, const Split3 &a, const Split3 &b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
}

}  // namespace

// dw[i] = sum_p part[p][i], i < count (mlp_gemm.hip): the deterministic reduction of the
// per-workgroup partial weight gradients
int mlp_reduce_partials(int count, int parts, const float *part, float *out, hipStream_t stream);
// The same for a WEIGHT gradient, whose only reader is the optimizer's gradient packing: while
// mlp_defer_weight_reductions(1) is in force the reduction is queued instead of launched, and
// mlp_flush_weight_reductions() runs every queued one in ONE launch (a backward pass has ~30 of
// them at 3-6 us each).  `part` must stay allocated until the flush.
int mlp_reduce_weight_partials(int count, int parts, const float *part, float *out, hipStream_t stream);
// the pooled 128 -> 256 layer's forward as a persistent T-form kernel (mlp_pool_fwd256.hip); y may be NULL
int mlp_pool_fwd256_supported(int b, int m, int k, int r, int ns, const float *w, const float *x);
int mlp_pool_fwd256_launch(int b, int r, int ns, const float *w, const float *x, const float *scale,
                           const float *shift, const float *gamma, float *y, float *pairs, float *ext,
                           hipStream_t stream);
// the (128, 128) layers the same way (fwd128_kernel): statistics as 64-column pairs; ns = 0: not pooled
int mlp_fwd128_enabled_for(int b, int m, int k, int r);
int mlp_fwd128_launch(int b, int r, int ns, int direct, const float *w, const float *x, const float *scale,
                      const float *shift, const float *gamma, float *y, float *pairs, float *ext,
                      hipStream_t stream);
