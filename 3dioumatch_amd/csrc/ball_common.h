// 3dioumatch_amd/csrc/ball_common.h -- the brute-force ball query of ONE wavefront, shared by
// the brute-force tier (pn2_ball_group.hip) and the cell-list tier's overflow path
// (pn2_ball_grid.hip).  Semantics: ball_query_gpu.cu:14-49, SURVEY App. A.3.
#pragma once
#include "common.h"

// One wave scans the cloud 64 consecutive points at a time for QW wave-uniform centroids and
// writes their rows: the first nsample indices (ascending) with d2 < radius2, the tail padded
// with the first hit, all zeros when there is none.  Hits are compacted in index order with
// a ballot + prefix popcount; a full row stops being tested (wave-uniform early exit).
template <int QW>
__device__ __forceinline__ void ball_query_wave_scan(const float *__restrict__ pts, int n,
                                                     const float *__restrict__ ctr, int live_q,
                                                     float radius2, int nsample,
                                                     int *__restrict__ rows) {
  const int lane = lane_id();
  float cx[QW], cy[QW], cz[QW];
  int cnt[QW], first[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    const bool live = q < live_q;
    cx[q] = live ? ctr[q * 3 + 0] : 0.f;
    cy[q] = live ? ctr[q * 3 + 1] : 0.f;
    cz[q] = live ? ctr[q * 3 + 2] : 0.f;
    cnt[q] = live ? 0 : nsample;  // dead slots count as already full
    first[q] = 0;
  }
  for (int base = 0; base < n; base += kWave) {
    bool any_open = false;
#pragma unroll
    for (int q = 0; q < QW; ++q) any_open |= cnt[q] < nsample;
    if (!any_open) break;  // wave-uniform
    const int k = base + lane;
    const bool valid = k < n;
    const float x = valid ? pts[k * 3 + 0] : 0.f;
    const float y = valid ? pts[k * 3 + 1] : 0.f;
    const float z = valid ? pts[k * 3 + 2] : 0.f;
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      if (cnt[q] < nsample) {  // wave-uniform
        const float d2 = sqdist3(cx[q], cy[q], cz[q], x, y, z);
        const bool hit = valid && d2 < radius2;
        const unsigned long long mask = __ballot(hit);
        if (mask) {
          if (cnt[q] == 0) first[q] = base + __builtin_ctzll(mask);
          const int slot = cnt[q] + mask_rank(mask);
          if (hit && slot < nsample) rows[(size_t)q * nsample + slot] = k;
          cnt[q] += __popcll(mask);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    if (q < live_q) {
      const int have = cnt[q] < nsample ? cnt[q] : nsample;
      for (int s = have + lane; s < nsample; s += kWave) rows[(size_t)q * nsample + s] = first[q];
    }
  }
}
