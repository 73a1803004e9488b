// 3dioumatch_amd/csrc/query_desc.h -- per-centroid DESCRIPTORS of the cell-list ball query
// (pn2_ball_grid.hip): for each centroid its coordinates and the nine row ranges of the cell lists
// it has to test (ball_query_gpu.cu:24-47 scans the whole cloud instead).
//
// Why a separate object: measured (profiles/r3_pair_*), the query kernel is bound by its chain of
// DEPENDENT memory round trips per centroid (~2000 cycles each under load: centroid -> CSR offsets
// -> candidate rows -> ... -> feature gather), not by bandwidth or instruction issue.  The
// descriptors take the first two links out of the chain: they are computed where the centroids are
// born -- the tail of the furthest-point-sampling kernel of the same set-abstraction layer, which
// has also just built the cell lists -- or by grid_desc_kernel.
//
// Layout, kDescInts ints per centroid:
//   [0..2]  x, y, z (bit patterns)   [3] 1 = a row is longer than 64 records or the centroid
//   sits at the lattice seam next to an occupied wrapped cell (the query takes its general path)
//   [4..12] first record of rows 0..8   [13..21] min(row length, 64)   [22, 23] unused
#pragma once
#include "common.h"
#include "grid_common.h"

namespace grid {

constexpr int kDescInts = 24;
constexpr int kDescThreads = 1024;  // lanes that call desc_build together

// All kDescThreads lanes of a workgroup call it; `st` = the cloud's CSR offsets (visible to this
// workgroup), xyz_of(j, x, y, z) = centroid j.
template <class XyzOf>
__device__ __forceinline__ void desc_build(int m, float inv_side, XyzOf xyz_of,
                                           const int *__restrict__ st, int *__restrict__ desc) {
  for (int j = (int)threadIdx.x; j < m; j += kDescThreads) {
    float x, y, z;
    xyz_of(j, x, y, z);
    const int gx = cell_coord(x, inv_side) & (kG - 1);
    const int gy = cell_coord(y, inv_side), gz = cell_coord(z, inv_side);
    const int xa = gx > 0 ? gx - 1 : 0, xb = gx < kG - 1 ? gx + 1 : kG - 1;
    const bool seam = gx == 0 || gx == kG - 1;
    int s0[9], s1[9], w0[9], w1[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {  // all loads first
      const int rowbase = (((gz + r / 3 - 1) & (kG - 1)) * kG + ((gy + r % 3 - 1) & (kG - 1))) * kG;
      s0[r] = st[rowbase + xa];
      s1[r] = st[rowbase + xb + 1];
      const int wc = rowbase + (gx == 0 ? kG - 1 : 0);
      w0[r] = seam ? st[wc] : 0;
      w1[r] = seam ? st[wc + 1] : 0;
    }
    int *d = desc + (size_t)j * kDescInts;
    int slow = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int len = s1[r] - s0[r];
      slow |= (len > kWave || w1[r] > w0[r]) ? 1 : 0;
      d[4 + r] = s0[r];
      d[13 + r] = len < kWave ? len : kWave;
    }
    d[0] = __builtin_bit_cast(int, x);
    d[1] = __builtin_bit_cast(int, y);
    d[2] = __builtin_bit_cast(int, z);
    d[3] = slow;
    d[22] = 0;
    d[23] = 0;
  }
}

}  // namespace grid
