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
constexpr int kChunks = 32;             // chunks a cloud is split into by the first build pass
constexpr int kSegOff = kG + 1;         // layer offsets per chunk (+ total)
constexpr int kGridMaxPoints = kChunks * 4096;  // the first pass keeps a chunk in registers

__host__ __device__ inline int grid_chunk_points(int n) {
  return (((n + kChunks - 1) / kChunks) + 3) & ~3;
}

struct GridWs {
  int *start;     // [b][kStartStride]
  int *segoff;    // [b][kChunks][kSegOff]        (scratch of the two-pass build)
  float4 *rec;    // [b][n]  records in cell order
  float4 *seg;    // [b][kChunks][chunk_pts]      (scratch of the two-pass build)
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

}  // namespace grid
