// 3dioumatch_amd/csrc/tile_plan.h -- the query plan of the tile form of the ball query
// (pn2_ball_tile.hip): centroids counting-sorted by 2x2x2-cell tile of the lattice of
// grid_common.h.  Built by grid_plan_kernel, or by the tail of the furthest-point-sampling kernel
// of the same set-abstraction layer (pn2_fps_bucket.hip), which has just picked the centroids.
#pragma once
#include "common.h"
#include "grid_common.h"

namespace grid {

constexpr int kTilesPerAxis = kG / 2;
constexpr int kTiles = kTilesPerAxis * kTilesPerAxis * kTilesPerAxis;  // 4096 per cloud
constexpr int kPlanThreads = 1024;       // lanes that call plan_build together

__device__ __forceinline__ int tile_of_cell(int gx, int gy, int gz) {
  return ((gz >> 1) * kTilesPerAxis + (gy >> 1)) * kTilesPerAxis + (gx >> 1);
}

// inclusive prefix sum over the 64 lanes with DPP row shifts / row broadcasts
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_t(int v) {
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_scan_incl(int v) {
  v = dpp_add_t<0x111, 0xf>(v);
  v = dpp_add_t<0x112, 0xf>(v);
  v = dpp_add_t<0x114, 0xf>(v);
  v = dpp_add_t<0x118, 0xf>(v);
  v = dpp_add_t<0x142, 0xa>(v);
  v = dpp_add_t<0x143, 0xc>(v);
  return v;
}

// ---- the plan: centroids counting-sorted by tile ------------------------------------------------
// plan layout per cloud (ints): [0] number of occupied tiles, [1..3] unused,
//   [4, 4 + m)        order: centroid ids, grouped by tile
//   [4 + m, 4 + 3m)   per occupied tile k: word 2k = tile id | count << 12, word 2k+1 = first
//                     position in `order`
// `hist` : kTiles ints of LDS (zeroed here), `scr` : 2 * 16 ints of LDS.  All 1024 lanes call it.
template <class XyzOf>
__device__ __forceinline__ void plan_build(int m, float inv_side, XyzOf xyz_of, int *hist,
                                           int *scr, int *__restrict__ plan) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
  for (int t = tid; t < kTiles; t += kPlanThreads) hist[t] = 0;
  __syncthreads();
  for (int j = tid; j < m; j += kPlanThreads) {
    float x, y, z;
    xyz_of(j, x, y, z);
    const int t = tile_of_cell(cell_coord(x, inv_side) & (kG - 1), cell_coord(y, inv_side) & (kG - 1),
                               cell_coord(z, inv_side) & (kG - 1));
    atomicAdd(&hist[t], 1);
  }
  __syncthreads();
  {  // exclusive scans of the counts and of the occupied flags: 4 consecutive tiles per lane
    int c[4], sum = 0, occ = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c[q] = hist[tid * 4 + q];
      sum += c[q];
      occ += c[q] > 0 ? 1 : 0;
    }
    const int isum = wave_scan_incl(sum), iocc = wave_scan_incl(occ);
    if (lane == kWave - 1) { scr[w] = isum; scr[16 + w] = iocc; }
    __syncthreads();
    int run = isum - sum, k = iocc - occ;
    for (int q = 0; q < w; ++q) { run += scr[q]; k += scr[16 + q]; }
    if (tid == kPlanThreads - 1) plan[0] = k + occ;
    int *tr = plan + 4 + m;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      hist[tid * 4 + q] = run;  // scatter cursor of the tile
      if (c[q] > 0) {
        tr[2 * k] = (tid * 4 + q) | (c[q] << 12);
        tr[2 * k + 1] = run;
        ++k;
      }
      run += c[q];
    }
  }
  __syncthreads();
  for (int j = tid; j < m; j += kPlanThreads) {
    float x, y, z;
    xyz_of(j, x, y, z);
    const int t = tile_of_cell(cell_coord(x, inv_side) & (kG - 1), cell_coord(y, inv_side) & (kG - 1),
                               cell_coord(z, inv_side) & (kG - 1));
    plan[4 + atomicAdd(&hist[t], 1)] = j;
  }
}


}  // namespace grid
