// 3dioumatch_amd/csrc/grid_common.h -- the cell lists shared by the ball-query tier
// (pn2_ball_grid.hip builds and queries them) and the bucketed furthest point sampling
// (pn2_fps_bucket.hip can leave them behind as a by-product of reading the cloud anyway).
//
// Lattice: cells of side 1.001 * radius, 32^3 PERIODIC (cell = floor(p / side) mod 32 per axis --
// no bounding-box pass; far-apart cells may alias, which only adds candidates that the exact
// distance test rejects).  Cell id = (z * 32 + y) * 32 + x.  Storage per cloud: CSR offsets
// `start[cell]` (start[kCells] = n) and the records (x, y, z, original index) in cell order.
#pragma once
#include "common.h"

namespace grid {

constexpr int kG = 32;                  // lattice cells per axis (periodic)
constexpr int kLayerCells = kG * kG;    // cells of one z-layer
constexpr int kCells = kG * kG * kG;
constexpr int kStartStride = kCells + 32;  // ints per cloud in `start` (start[kCells] = n)
// start[kOrderFor]: the number of centroids `order` of this cloud is a permutation for, 0 = none.
// Written 0 by every builder of the lists, m by the sampling kernel that also knows the centroids.
constexpr int kOrderFor = kCells + 1;
constexpr int kChunks = 32;             // chunks a cloud is split into by the first build pass
constexpr int kSegOff = kG + 1;         // layer offsets per chunk (+ total)
constexpr int kGridMaxPoints = kChunks * 4096;  // the first pass keeps a chunk in registers
// Query plans (below): 32 words per centroid, room for n / 8 centroids per cloud
constexpr int kPlanWords = 32;
__host__ __device__ inline int grid_plan_capacity(int n) { return n / 8; }

__host__ __device__ inline int grid_chunk_points(int n) {
  return (((n + kChunks - 1) / kChunks) + 3) & ~3;
}

struct GridWs {
  int *start;     // [b][kStartStride]
  int *segoff;    // [b][kChunks][kSegOff]        (scratch of the two-pass build)
  float4 *rec;    // [b][n]  records in cell order
  float4 *seg;    // [b][kChunks][chunk_pts]      (scratch of the two-pass build)
  // [b][n] launch order of the centroids sampled from this cloud: longest query first (the query
  // kernel's workgroup jj answers centroid order[jj]); keys of the counting sort behind it
  int *order;
  int *order_key;
  // [b][n / 8][32] query plans of the centroids sampled from this cloud, in launch order
  unsigned *plan;
  size_t bytes;
};

inline GridWs grid_ws_layout(void *base, int b, int n) {
  GridWs w;
  char *p = reinterpret_cast<char *>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { char *q = p + off; off += (bytes + 255) & ~(size_t)255; return q; };
  w.start = reinterpret_cast<int *>(take(sizeof(int) * (size_t)b * kStartStride));
  w.segoff = reinterpret_cast<int *>(take(sizeof(int) * (size_t)b * kChunks * kSegOff));
  w.rec = reinterpret_cast<float4 *>(take(sizeof(float4) * (size_t)b * n));
  w.seg = reinterpret_cast<float4 *>(take(sizeof(float4) * (size_t)b * kChunks * grid_chunk_points(n)));
  w.order = reinterpret_cast<int *>(take(sizeof(int) * (size_t)b * n));
  w.order_key = reinterpret_cast<int *>(take(sizeof(int) * (size_t)b * n));
  w.plan = reinterpret_cast<unsigned *>(take(sizeof(unsigned) * kPlanWords * (size_t)b * grid_plan_capacity(n)));
  w.bytes = off;
  return w;
}

inline float grid_inv_side(float radius) { return 1.0f / (radius * 1.001f); }

__device__ __forceinline__ int cell_coord(float v, float inv_side) {
  return (int)floorf(v * inv_side);
}

__device__ __forceinline__ int cell_id(float x, float y, float z, float inv_side) {
  return (((cell_coord(z, inv_side) & (kG - 1)) * kG + (cell_coord(y, inv_side) & (kG - 1))) * kG) +
         (cell_coord(x, inv_side) & (kG - 1));
}

// Cost class of a centroid's query (what grid_query_kernel will do for it, from the row lengths
// alone): 0 = every one of its nine x-rows is shorter than a wave (the single-load path), otherwise the
// number of 64-record chunks its general path sweeps, capped at 63.
__device__ inline int query_cost_class(const int *st, float cx, float cy, float cz, float inv_side) {
  const int gx = cell_coord(cx, inv_side) & (kG - 1);
  const int gy = cell_coord(cy, inv_side), gz = cell_coord(cz, inv_side);
  const int xa = gx > 0 ? gx - 1 : 0, xb = gx < kG - 1 ? gx + 1 : kG - 1;
  const bool seam = gx == 0 || gx == kG - 1;
  bool fast = true;
  int chunks = 0;
  for (int r = 0; r < 9; ++r) {
    const int rz = r / 3;
    const int rowbase = (((gz + rz - 1) & (kG - 1)) * kG + ((gy + (r - 3 * rz) - 1) & (kG - 1))) * kG;
    const int len = st[rowbase + xb + 1] - st[rowbase + xa];
    int lenw = 0;
    if (seam) {
      const int c = rowbase + (gx == 0 ? kG - 1 : 0);
      lenw = st[c + 1] - st[c];
    }
    fast = fast && len + lenw < kWave;
    chunks += ((len + kWave - 1) >> 6) + ((lenw + kWave - 1) >> 6);
  }
  return fast ? 0 : (chunks < 63 ? (chunks > 0 ? chunks : 1) : 63);
}

// The QUERY PLAN of a centroid: everything grid_query_kernel's wave needs before its row loads, as
// ONE 64-byte scalar load (a second half for the centroids that need it) -- written by the kernel
// that knows both the centroids and the lists (the sampling kernel, next to the launch order;
// entry jj of a cloud = the centroid its workgroup jj answers).  Without it the wave walks
// order -> centroid -> cell coordinates -> row offsets: three dependent trips to the L2 and ~45
// vector + ~90 scalar instructions.
// A plan is a sequence of PASSES of nine loads of at most 63 records each (a lane mask of 64
// lanes does not come out of s_bfm_b64), all nine in flight before the first test:
//   words 0..8 : byte offset of row r's first record in the cloud's record array
//   word  9    : pass 0, lengths of loads 0..4, six bits each
//   word 10    : pass 0, lengths of loads 5..8, six bits each; bits 24..: the plan's kind --
//                0 = pass 0 is the whole neighbourhood (every row shorter than 64 records, no
//                wrapped cell); 2 = more passes follow; 1 = the wave computes its rows itself
//   word 11    : centroid j | m << 16 (the m it was made for: a wave that finds another m here
//                does not use the plan)
//   words 12..14: the centroid's coordinates
//   word 15    : the query's cost class | further passes << 8 | (one of them is the wrapped cells) << 16
// Further passes (dense clouds: rows of 64 records and more; walls at the lattice seam with
// points on both sides: the cell that wraps around): pass t = 1, 2, ... reads records
// 63 t .. 63 t + 62 of every row -- the same nine offsets + 1008 t bytes -- and its two words of
// lengths are words 16 + 2 (t - 1), ...; with wrapped cells, words 16..24 are THEIR nine offsets,
// the length pairs start at word 25 and the last pass is theirs.  Room: 8 further passes (rows of
// up to 567 records), 3 with wrapped cells; beyond that, or a wrapped cell of 64 records: kind 1.
constexpr int kPlanRecords = kWave - 1;  // records per load
__device__ inline void write_query_plan(unsigned *rec, const int *st, float cx, float cy, float cz,
                                        float inv_side, int j, int m, int cost) {
  const int gx = cell_coord(cx, inv_side) & (kG - 1);
  const int gy = cell_coord(cy, inv_side), gz = cell_coord(cz, inv_side);
  const int xa = gx > 0 ? gx - 1 : 0, xb = gx < kG - 1 ? gx + 1 : kG - 1;
  const bool seam = gx == 0 || gx == kG - 1;
  unsigned w[kPlanWords];
  for (int i = 0; i < kPlanWords; ++i) w[i] = 0u;
  int len[9], lenw[9], sw[9];
  int longest = 0, longest_w = 0;
  for (int r = 0; r < 9; ++r) {
    const int rz = r / 3;
    const int rowbase = (((gz + rz - 1) & (kG - 1)) * kG + ((gy + (r - 3 * rz) - 1) & (kG - 1))) * kG;
    const int s0 = st[rowbase + xa];
    len[r] = st[rowbase + xb + 1] - s0;
    sw[r] = 0; lenw[r] = 0;
    if (seam) {
      const int c = rowbase + (gx == 0 ? kG - 1 : 0);
      sw[r] = st[c];
      lenw[r] = st[c + 1] - sw[r];
    }
    w[r] = (unsigned)s0 * 16u;
    longest = len[r] > longest ? len[r] : longest;
    longest_w = lenw[r] > longest_w ? lenw[r] : longest_w;
  }
  const bool has_w = longest_w > 0;
  const int main_passes = longest > 0 ? (longest + kPlanRecords - 1) / kPlanRecords : 1;
  const int further = main_passes - 1 + (has_w ? 1 : 0);
  const bool ok = longest_w <= kPlanRecords && further <= (has_w ? 3 : 8);
  auto pack = [&](int word, int r, int l) {  // six bits per load, five loads in the first word
    w[word + (r < 5 ? 0 : 1)] |= (unsigned)l << (6 * (r < 5 ? r : r - 5));
  };
  auto clamp63 = [](int v) { return v < 0 ? 0 : (v < kPlanRecords ? v : kPlanRecords); };
  for (int r = 0; r < 9; ++r) pack(9, r, clamp63(len[r]));
  if (ok && further > 0) {
    const int pairs = has_w ? 25 : 16;
    for (int t = 1; t < main_passes; ++t)
      for (int r = 0; r < 9; ++r) pack(pairs + 2 * (t - 1), r, clamp63(len[r] - kPlanRecords * t));
    if (has_w)
      for (int r = 0; r < 9; ++r) {
        w[16 + r] = (unsigned)sw[r] * 16u;
        pack(pairs + 2 * (main_passes - 1), r, lenw[r]);
      }
  }
  w[10] |= (ok ? (further > 0 ? 2u : 0u) : 1u) << 24;
  w[11] = (unsigned)j | (unsigned)m << 16;
  w[12] = __builtin_bit_cast(unsigned, cx);
  w[13] = __builtin_bit_cast(unsigned, cy);
  w[14] = __builtin_bit_cast(unsigned, cz);
  w[15] = (unsigned)(cost & 0xff) | (ok ? (unsigned)further << 8 | (has_w ? 1u << 16 : 0u) : 0u);
  uint4 *o = reinterpret_cast<uint4 *>(rec);
  for (int i = 0; i < kPlanWords / 4; ++i) o[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

}  // namespace grid
