// 3dioumatch_amd/csrc/iou3d.hip -- rotated-box BEV overlap / IoU / 3-D IoU matrices and NMS
// for gfx950.
//
// Semantics: reference iou3d_nms_kernel.cu (K10 :249-262, K11 :264-278, K12 :280-324,
// K13 :341-385), host scan iou3d_nms.cpp:121-134, Python epilogue iou3d_nms_utils.py:60-79,
// CPU op iou3d_cpu.cpp:232-252.  Geometry: box_geom.h.
//
// Design:
//  * pair matrices: 256-lane workgroups own a 16x16 tile of the (Na,Nb) matrix; the 32 boxes
//    of the tile are pre-processed ONCE into LDS (rotated corners, cos/sin, inflated half
//    extents) -- the reference redoes that per pair and its cos/sin per corner test;
//  * the dynamically indexed polygon (<=16 vertices + polar angles) lives in LDS, laid out
//    [slot][lane] so the 64 lanes of a wave never collide on a bank (no scratch memory);
//  * far-apart / z-disjoint pairs exit before any trigonometry with the exact value 0;
//  * NMS: one LANE per (row, column) pair and a wave ballot to form each 64-bit mask word
//    (the reference gives each thread a whole row and loops 64 columns serially); the greedy
//    scan runs on the device in a single wavefront, so nms needs no device->host mask copy.
#include "common.h"
#include "box_geom.h"
#include <mutex>

namespace {

using boxgeom::BoxPre;

constexpr int kPolySlots = boxgeom::kMaxPoly * 3;  // x, y, angle per vertex

// polygon store in LDS: element (slot s, lane t) at base[s*256 + t]
struct LdsPoly {
  float *base;
  __device__ __forceinline__ float &x(int i) { return base[(i * 3 + 0) * 256]; }
  __device__ __forceinline__ float &y(int i) { return base[(i * 3 + 1) * 256]; }
  __device__ __forceinline__ float &a(int i) { return base[(i * 3 + 2) * 256]; }
};

struct HostPoly {
  float vx[boxgeom::kMaxPoly], vy[boxgeom::kMaxPoly], va[boxgeom::kMaxPoly];
  float &x(int i) { return vx[i]; }
  float &y(int i) { return vy[i]; }
  float &a(int i) { return va[i]; }
};

enum PairMode { kOverlap = 0, kIouBev = 1, kIou3d = 2 };

template <int MODE>
__device__ __forceinline__ float pair_value(const float *a, const float *b, const BoxPre &A,
                                            const BoxPre &B, LdsPoly &st) {
  if (MODE == kOverlap) return boxgeom::overlap_area(A, B, st);
  if (MODE == kIouBev) {  // iou_bev, :228-235
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    const float ov = boxgeom::overlap_area(A, B, st);
    return ov / fmaxf(sa + sb - ov, 1e-8f);
  }
  // boxes_iou3d_gpu epilogue, iou3d_nms_utils.py:60-79
  const float a_max = a[2] + a[5] / 2, a_min = a[2] - a[5] / 2;
  const float b_max = b[2] + b[5] / 2, b_min = b[2] - b[5] / 2;
  const float max_of_min = a_min > b_min ? a_min : b_min;
  const float min_of_max = a_max < b_max ? a_max : b_max;
  float h = min_of_max - max_of_min;
  if (h < 0.f) h = 0.f;
  const float ov_bev = h > 0.f ? boxgeom::overlap_area(A, B, st) : 0.f;
  const float ov3d = ov_bev * h;
  const float vol_a = a[3] * a[4] * a[5], vol_b = b[3] * b[4] * b[5];
  float den = vol_a + vol_b - ov3d;
  if (den < 1e-6f) den = 1e-6f;
  return ov3d / den;
}

// ans[i,j] for a 16x16 tile; threadIdx.x = 16*row + col
template <int MODE>
__global__ void __launch_bounds__(256)
pair_matrix_kernel(int na, const float *__restrict__ boxes_a, int nb,
                   const float *__restrict__ boxes_b, float *__restrict__ ans) {
  __shared__ float poly[kPolySlots * 256];
  __shared__ BoxPre pre[32];
  __shared__ float raw[32 * 7];
  const int tid = threadIdx.x;
  const int row0 = blockIdx.y * 16, col0 = blockIdx.x * 16;
  if (tid < 32) {
    const bool is_a = tid < 16;
    const int g = is_a ? row0 + tid : col0 + (tid - 16);
    const bool ok = is_a ? g < na : g < nb;
    const float *src = (is_a ? boxes_a : boxes_b) + (size_t)(ok ? g : 0) * 7;
    float bx[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) { bx[d] = src[d]; raw[tid * 7 + d] = bx[d]; }
    boxgeom::box_prepare(bx, pre[tid]);
  }
  __syncthreads();
  const int r = tid >> 4, cidx = tid & 15;
  const int gi = row0 + r, gj = col0 + cidx;
  if (gi >= na || gj >= nb) return;
  const BoxPre A = pre[r];
  const BoxPre B = pre[16 + cidx];
  LdsPoly st{poly + tid};
  ans[(size_t)gi * nb + gj] = pair_value<MODE>(raw + r * 7, raw + (16 + cidx) * 7, A, B, st);
}

// Per-scene best match: for prediction i of scene s, max_j iou3d(a[s][i], b[s][j]) and the first j
// attaining it -- the block-diagonal of the all-pairs matrix and the max/gather the reference
// applies to it (loss_helper_iou.py:106-111), without the (S-1)/S cross-scene pairs.
// Workgroup = 16 predictions of one scene; lane (row, col) walks the scene's boxes 16 at a time.
__global__ void __launch_bounds__(256)
scene_max_kernel(int na, int nb, const float *__restrict__ boxes_a,
                 const float *__restrict__ boxes_b, float *__restrict__ best_iou,
                 int *__restrict__ best_idx) {
  __shared__ float poly[kPolySlots * 256];
  __shared__ BoxPre pre[32];
  __shared__ float raw[32 * 7];
  const int tid = threadIdx.x;
  const int scene = blockIdx.y, row0 = blockIdx.x * 16;
  const float *sa = boxes_a + (size_t)scene * na * 7;
  const float *sb = boxes_b + (size_t)scene * nb * 7;
  const int r = tid >> 4, cidx = tid & 15;
  const int gi = row0 + r;
  float best = -1.f;  // every IoU is >= 0, so column 0 wins an all-zero row
  int arg = 0;
  LdsPoly st{poly + tid};
  for (int col0 = 0; col0 < nb; col0 += 16) {
    __syncthreads();
    if (tid < 32) {
      const bool is_a = tid < 16;
      const int g = is_a ? row0 + tid : col0 + (tid - 16);
      const bool ok = is_a ? g < na : g < nb;
      const float *src = (is_a ? sa : sb) + (size_t)(ok ? g : 0) * 7;
      if (!is_a || col0 == 0) {
        float bx[7];
#pragma unroll
        for (int d = 0; d < 7; ++d) { bx[d] = src[d]; raw[tid * 7 + d] = bx[d]; }
        boxgeom::box_prepare(bx, pre[tid]);
      }
    }
    __syncthreads();
    const int gj = col0 + cidx;
    if (gi < na && gj < nb) {
      const float v = pair_value<kIou3d>(raw + r * 7, raw + (16 + cidx) * 7, pre[r],
                                         pre[16 + cidx], st);
      if (v > best) { best = v; arg = gj; }  // strict: the earlier column keeps a tie
    }
  }
  // the 16 lanes of a row are consecutive: butterfly over them, (value desc, index asc)
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(best, off, 16);
    const int oa = __shfl_xor(arg, off, 16);
    if (ov > best || (ov == best && oa < arg)) { best = ov; arg = oa; }
  }
  if (cidx == 0 && gi < na) {
    best_iou[(size_t)scene * na + gi] = best < 0.f ? 0.f : best;
    best_idx[(size_t)scene * na + gi] = arg;
  }
}

// iou_bev_3D (:237-247) / iou_normal (:327-338) as used by the NMS kernels
template <bool NORMAL>
__device__ __forceinline__ float nms_iou(const float *a, const float *b, const BoxPre &A,
                                         const BoxPre &B, LdsPoly &st) {
  if (NORMAL) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2);
    const float right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2);
    const float bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float inter = width * height;
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    return inter / fmaxf(sa + sb - inter, 1e-8f);
  }
  const float sa = a[3] * a[4] * a[5], sb = b[3] * b[4] * b[5];
  const float top = fmaxf(a[2] - a[5] / 2, b[2] - b[5] / 2);
  const float bottom = fminf(a[2] + a[5] / 2, b[2] + b[5] / 2);
  const float height = fmaxf(bottom - top, 0.f);
  const float ov = height > 0.f ? boxgeom::overlap_area(A, B, st) : 0.f;
  const float s_overlap = ov * height;
  return s_overlap / fmaxf(sa + sb - s_overlap, 1e-8f);
}

// One workgroup = kMaskRows rows x one 64-column tile; each wave walks kMaskRows / 4 rows, its 64
// lanes are the 64 columns; the mask word is the wave ballot.  Lower-triangle tiles are skipped
// unless FULL (the scan never reads them, iou3d_nms.cpp:129-131).  (Four rows per workgroup instead
// of 16 -- four times the workgroups for the 1024-box case -- was measured in round 6: 88 us
// against 76; profiles/r6_ops_time.json.)
constexpr int kMaskRows = 16;
template <bool NORMAL>
__global__ void __launch_bounds__(256)
nms_mask_kernel(int n, float thresh, int full, const float *__restrict__ boxes,
                unsigned long long *__restrict__ mask) {
  __shared__ float poly[kPolySlots * 256];
  __shared__ BoxPre pre[64 + kMaskRows];
  __shared__ float raw[(64 + kMaskRows) * 7];
  const int tid = threadIdx.x;
  const int cb = blockIdx.x;
  const int row0 = blockIdx.y * kMaskRows;
  const int rb = row0 >> 6;
  if (!full && cb < rb) return;  // uniform for the whole workgroup
  const int col_blocks = (n + 63) / 64;
  if (tid < 64 + kMaskRows) {
    const int g = tid < 64 ? cb * 64 + tid : row0 + (tid - 64);
    const float *src = boxes + (size_t)(g < n ? g : 0) * 7;
    float bx[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) { bx[d] = src[d]; raw[tid * 7 + d] = bx[d]; }
    if (!NORMAL) boxgeom::box_prepare(bx, pre[tid]);
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int gc = cb * 64 + lane;
  LdsPoly st{poly + tid};
  for (int it = 0; it < kMaskRows / 4; ++it) {
    const int rl = wave * (kMaskRows / 4) + it;
    const int gr = row0 + rl;
    if (gr >= n) break;  // wave-uniform
    bool bit = false;
    const bool test = gc < n && !(rb == cb && gc <= gr);
    if (test)
      bit = nms_iou<NORMAL>(raw + (64 + rl) * 7, raw + lane * 7, pre[64 + rl], pre[lane], st) >
            thresh;
    const unsigned long long word = __ballot(bit);
    if (lane == 0) mask[(size_t)gr * col_blocks + cb] = word;
  }
}

// Greedy scan, iou3d_nms.cpp:121-134, in one wavefront, BLOCK BY BLOCK of 64 boxes.
// The reference walks the boxes one at a time and ORs a kept box's mask row into `remv` -- as a
// device loop that is one dependent trip to memory per kept box (0.5 us each: 330 us for 660
// survivors of 1024, against 35 us for the mask itself).  Here, for block nb:
//   * lane l holds the DIAGONAL word of row nb * 64 + l (who of the same block it suppresses).
//     The block's survivors are the unique fixed point of K = candidates & ~OR_{i in K} diag_i
//     (the matrix is strictly upper triangular: box i depends on lower numbers only), reached by
//     iterating from K = candidates -- the lowest undecided box is decided in every round, in
//     practice a handful of rounds of one wave-wide OR each (DPP), no memory, no 64-step loop;
//   * the survivors' rows are ORed into remv for the later blocks with lane = column word: up to
//     1024 boxes the whole mask sits in LDS (copied in one trip: all loads in flight) and a row is
//     one LDS read; beyond, eight independent row loads from memory in flight at a time and the
//     next block's diagonal words requested before this block's rounds start.
// Same keep list, same order (ascending box number) as the reference's scan.
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
  auto step = [](unsigned x, auto ctrl, auto row_mask) {
    return x | (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, decltype(row_mask)::value,
                                                     0xf, false);
  };
  v = step(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
  v = step(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
  v = step(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
  v = step(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8
  v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15
  v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31
  return (unsigned)__builtin_amdgcn_readlane((int)v, kWave - 1);
}

// LDSM: the whole mask (n x col_blocks words, <= 128 KB: n <= 1024) is copied into LDS first --
// every load in flight at once, one trip to memory for the whole scan -- and remv lives in a
// register (lane c holds word c).  Otherwise: any size, rows from memory.
template <bool LDSM>
__global__ void __launch_bounds__(LDSM ? 1024 : 64)
nms_scan_kernel(const unsigned long long *__restrict__ mask, int n, int col_blocks,
                long long *__restrict__ keep, int *__restrict__ num_out) {
  extern __shared__ unsigned long long lds_words[];  // LDSM: the mask; else remv
  const int lane = threadIdx.x;
  unsigned long long *remv = lds_words;
  if (LDSM) {  // 16 waves copy (<= 16 words per lane, one trip to memory); wave 0 scans
    const int words = n * col_blocks;
    unsigned long long v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = u * 1024 + (int)threadIdx.x;
      v[u] = t < words ? mask[t] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = u * 1024 + (int)threadIdx.x;
      if (t < words) lds_words[t] = v[u];
    }
  } else {
    for (int l = lane; l < col_blocks; l += kWave) remv[l] = 0ull;
  }
  __syncthreads();
  if (LDSM && threadIdx.x >= kWave) return;
  auto word_at = [&](int r, int c) -> unsigned long long {
    return LDSM ? lds_words[r * col_blocks + c] : mask[(size_t)r * col_blocks + c];
  };
  auto diagonal = [&](int nb) -> unsigned long long {
    const int r = nb * kWave + lane;
    return nb < col_blocks && r < n ? word_at(r, nb) : 0ull;
  };
  auto uniform64 = [](unsigned long long v, int from) -> unsigned long long {
    return (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), from) << 32 |
           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, from);
  };
  int num = 0;
  unsigned long long removed = 0ull;  // LDSM: lane c holds remv[c]
  unsigned long long diag = diagonal(0);
  for (int nb = 0; nb < col_blocks; ++nb) {
    const unsigned long long diag_next = LDSM ? 0ull : diagonal(nb + 1);
    // removed so far (wave-uniform); boxes past the end count as removed
    unsigned long long w = LDSM ? uniform64(removed, nb) : uniform64(remv[nb], 0);
    const int in_block = n - nb * kWave;
    if (in_block < kWave) w |= ~0ull << in_block;
    const unsigned long long cand = ~w;
    unsigned long long kept = cand;
    for (;;) {  // wave-uniform
      const bool in = (kept >> lane) & 1ull;
      const unsigned lo = wave_or_u32(in ? (unsigned)diag : 0u), hi = wave_or_u32(in ? (unsigned)(diag >> 32) : 0u);
      const unsigned long long next = cand & ~((unsigned long long)hi << 32 | lo);
      if (next == kept) break;
      kept = next;
    }
    if ((kept >> lane) & 1ull)
      keep[num + __popcll(kept & ((1ull << lane) - 1ull))] = (long long)nb * kWave + lane;
    num += __popcll(kept);
    if (LDSM) {
      // the survivors' rows into remv: lane = column word (col_blocks <= 16 < 64 lanes)
      const bool mine = lane > nb && lane < col_blocks;
      const unsigned long long *col = lds_words + (mine ? lane : 0);
      unsigned long long acc = 0ull;
      unsigned long long k = kept;
      while (k) {  // wave-uniform: one LDS read per survivor, all independent
        const int i = __builtin_ctzll(k);
        k &= k - 1ull;
        acc |= col[(nb * kWave + i) * col_blocks];
      }
      if (mine) removed |= acc;
      diag = diagonal(nb + 1);
    } else {
      // the survivors' rows into remv[nb + 1 ..]: lane = column word
      for (int c0 = 0; c0 < col_blocks; c0 += kWave) {
        const int c = c0 + lane;
        if (c0 + kWave - 1 <= nb) continue;  // wave-uniform: nothing right of the diagonal here
        const bool mine = c > nb && c < col_blocks;
        const unsigned long long *col = mask + (mine ? c : 0);
        unsigned long long acc = 0ull;
        unsigned long long k = kept;
        while (k) {  // wave-uniform
          unsigned long long v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {  // branch-free: eight loads in flight
            const bool have = k != 0ull;
            const int i = have ? __builtin_ctzll(k) : 0;
            k &= k - 1ull;
            const unsigned long long x = col[(size_t)(nb * kWave + i) * col_blocks];
            v[u] = have ? x : 0ull;
          }
          acc |= (v[0] | v[1]) | (v[2] | v[3]) | ((v[4] | v[5]) | (v[6] | v[7]));
        }
        if (mine) remv[c] |= acc;
      }
      __syncthreads();
      diag = diag_next;
    }
  }
  if (lane == 0) *num_out = num;
}

}  // namespace

#define IOU3D_API extern "C" __attribute__((visibility("default")))

template <int MODE>
static int launch_pairs(int na, const float *a, int nb, const float *b, float *ans,
                        void *stream_) {
  if (na <= 0 || nb <= 0) return 0;
  dim3 grid(pn2_ceil_div(nb, 16), pn2_ceil_div(na, 16));
  hipLaunchKernelGGL(pair_matrix_kernel<MODE>, grid, dim3(256), 0, (hipStream_t)stream_, na, a,
                     nb, b, ans);
  return pn2_launch_status();
}

IOU3D_API int iou3d_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b,
                                      const float *boxes_b, float *ans, void *stream) {
  return launch_pairs<kOverlap>(num_a, boxes_a, num_b, boxes_b, ans, stream);
}

IOU3D_API int iou3d_boxes_iou_bev(int num_a, const float *boxes_a, int num_b,
                                  const float *boxes_b, float *ans, void *stream) {
  return launch_pairs<kIouBev>(num_a, boxes_a, num_b, boxes_b, ans, stream);
}

IOU3D_API int iou3d_boxes_iou3d(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                                float *ans, void *stream) {
  return launch_pairs<kIou3d>(num_a, boxes_a, num_b, boxes_b, ans, stream);
}

IOU3D_API int iou3d_scene_best_iou3d(int scenes, int num_a, const float *boxes_a, int num_b,
                                     const float *boxes_b, float *best_iou, int *best_idx,
                                     void *stream) {
  if (scenes <= 0 || num_a <= 0) return 0;
  if (num_b <= 0) {
    int rc = pn2_zero_async(best_iou, sizeof(float) * (size_t)scenes * num_a, (hipStream_t)stream);
    if (rc == 0) rc = pn2_zero_async(best_idx, sizeof(int) * (size_t)scenes * num_a, (hipStream_t)stream);
    return rc;
  }
  hipLaunchKernelGGL(scene_max_kernel, dim3(pn2_ceil_div(num_a, 16), scenes), dim3(256), 0,
                     (hipStream_t)stream, num_a, num_b, boxes_a, boxes_b, best_iou, best_idx);
  return pn2_launch_status();
}

static int launch_mask(const float *boxes, unsigned long long *mask, int n, float thresh,
                       bool normal, int full, hipStream_t stream) {
  if (n <= 0) return 0;
  dim3 grid((n + 63) / 64, pn2_ceil_div(n, kMaskRows));
  if (normal)
    hipLaunchKernelGGL(nms_mask_kernel<true>, grid, dim3(256), 0, stream, n, thresh, full, boxes,
                       mask);
  else
    hipLaunchKernelGGL(nms_mask_kernel<false>, grid, dim3(256), 0, stream, n, thresh, full, boxes,
                       mask);
  return pn2_launch_status();
}

IOU3D_API int iou3d_nms_mask(const float *boxes, unsigned long long *mask, int boxes_num,
                             float thresh, void *stream) {
  return launch_mask(boxes, mask, boxes_num, thresh, false, 1, (hipStream_t)stream);
}

IOU3D_API int iou3d_nms_normal_mask(const float *boxes, unsigned long long *mask, int boxes_num,
                                    float thresh, void *stream) {
  return launch_mask(boxes, mask, boxes_num, thresh, true, 1, (hipStream_t)stream);
}

IOU3D_API int iou3d_nms(const float *boxes, int boxes_num, float thresh, int normal,
                        unsigned long long *mask_ws, long long *keep_dev, int *num_out_dev,
                        void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (boxes_num <= 0) return pn2_zero_async(num_out_dev, sizeof(int), stream);
  int rc = launch_mask(boxes, mask_ws, boxes_num, thresh, normal != 0, 0, stream);
  if (rc != 0) return rc;
  const int col_blocks = (boxes_num + 63) / 64;
  if (col_blocks <= 16) {  // the mask fits the LDS
    static std::mutex mu;
    static bool attr_set = false;
    {
      std::lock_guard<std::mutex> lock(mu);
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(nms_scan_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 16 * 8);
        attr_set = true;
      }
    }
    hipLaunchKernelGGL(nms_scan_kernel<true>, dim3(1), dim3(1024),
                       sizeof(unsigned long long) * (size_t)boxes_num * col_blocks, stream, mask_ws, boxes_num,
                       col_blocks, keep_dev, num_out_dev);
  } else {
    hipLaunchKernelGGL(nms_scan_kernel<false>, dim3(1), dim3(64), sizeof(unsigned long long) * col_blocks,
                       stream, mask_ws, boxes_num, col_blocks, keep_dev, num_out_dev);
  }
  return pn2_launch_status();
}

// The reference's only native CPU op on this path (iou3d_cpu.cpp:232-252): host pointers,
// one thread, same arithmetic.  This is an API-mandated CPU operator, not a fallback for
// the device entry points above.
IOU3D_API int iou3d_boxes_iou_bev_cpu(int num_a, const float *boxes_a, int num_b,
                                      const float *boxes_b, float *ans) {
  if (num_a <= 0 || num_b <= 0) return 0;
  BoxPre *pb = new BoxPre[num_b];
  for (int j = 0; j < num_b; ++j) boxgeom::box_prepare(boxes_b + (size_t)j * 7, pb[j]);
  HostPoly st;
  for (int i = 0; i < num_a; ++i) {
    const float *a = boxes_a + (size_t)i * 7;
    BoxPre A;
    boxgeom::box_prepare(a, A);
    const float sa = a[3] * a[4];
    for (int j = 0; j < num_b; ++j) {
      const float *b = boxes_b + (size_t)j * 7;
      const float sb = b[3] * b[4];
      const float ov = boxgeom::overlap_area(A, pb[j], st);
      ans[(size_t)i * num_b + j] = ov / fmaxf(sa + sb - ov, 1e-8f);
    }
  }
  delete[] pb;
  return 0;
}
