// 3dioumatch_amd/csrc/pn2_ball_grid.hip -- cell-list tier of ball_query for large clouds.
//
// Semantics: ball_query_gpu.cu:14-49 (first nsample indices in ascending order with
// d2 < r^2, tail padded with the first hit, zero row without a hit); SURVEY App. A.3.
//
// The brute-force formulation needs B*m*N distance tests (6.6e8 at B=8, N=40000, m=2048:
// VALU-bound, ~0.4 ms) for 8 MB of compulsory traffic.  This tier brings the work down to the
// ~27 cells around each centroid:
//
//   build  : every point goes into a cell of side 1.001*r of a 32^3 PERIODIC lattice
//            (cell = floor(p / side) mod 32 per axis -- no bounding box pass, far-apart cells
//            may alias, which only adds candidates that the exact distance test rejects).
//            Cells are fixed-capacity slot arrays of (x, y, z, index); each workgroup owns one
//            z-layer and ranks its points with LDS atomics (no global atomics, no counter
//            memset); points beyond a cell's capacity go to a per-layer overflow list that the
//            queries of the neighbouring layers also scan (dense clumps stay exact and fast).
//   query  : one wavefront per centroid.  Lanes 0..26 fetch the 27 neighbour cell counts;
//            the nine x-rows of three cells are streamed 64 candidates at a time (all loads
//            issued before the first use), hits are compacted into an LDS list by ballot
//            rank, then the <=64 smallest indices are selected and ordered IN REGISTERS with a
//            64-lane bitonic network built from DPP row moves and permlane swaps (no LDS, no
//            scratch): chunk 0 is sorted ascending, every further chunk descending and folded
//            in with one min + a 6-stage bitonic merge.  Lane s then writes slot s of the row.
//            For 64 < nsample <= 128 a second pass collects the next 64 the same way.  Flagged
//            clouds, rows with more than kMaxHits hits and nsample > 128 fall back to
//            the brute-force scan of ball_common.h inside the same launch.
//
// The order in which atomics fill a cell is irrelevant: selection and ordering are by index.
#include "common.h"
#include "ball_common.h"

namespace {

constexpr int kG = 32;                  // lattice cells per axis (periodic)
constexpr int kCellsPerCloud = kG * kG * kG;
constexpr int kCap = 64;                // slots per cell
constexpr int kMaxHits = 384;           // LDS hit list per wave
constexpr int kRowSlots = 3 * kCap;     // candidates in one x-row of cells (<= 192)
constexpr int kRowPasses = kRowSlots / kWave;  // 3

constexpr int kOvfCap = 2048;           // overflow points kept per (cloud, z-layer)

// ints: cell counters, then per (cloud, slab): overflow flag and overflow-list length
__host__ __device__ inline size_t grid_cnt_bytes(int b) {
  return sizeof(int) * ((size_t)b * kCellsPerCloud + (size_t)((b * 64 + 63) / 64) * 64);
}

__device__ __forceinline__ int cell_coord(float v, float inv_side) {
  return (int)floorf(v * inv_side);
}

__device__ __forceinline__ int cell_index(int cx, int cy, int cz) {
  return ((cz & (kG - 1)) * kG + (cy & (kG - 1))) * kG + (cx & (kG - 1));
}

// Build without global atomics and without a memset: the lattice is cut into kSlabs slabs of
// kLayers z-layers; workgroup (slab, cloud) scans the WHOLE cloud (480 KB from L2, coalesced),
// keeps the points whose z-layer falls in its slab, ranks them with LDS atomics (its
// kLayers*1024 counters live in LDS) and writes their slots; finally it writes every counter of
// its slab, so all 32768 counters of the cloud are (re)written on every call.
// (Returning device-scope atomics execute at the memory side on this multi-XCD part:
// one per point cost 20 us for 320 000 points; this formulation costs a few us.)
constexpr int kSlabs = 32;                     // (grid_cnt_bytes reserves 32 flags per cloud)
constexpr int kLayers = kG / kSlabs;           // 1 z-layer per slab
constexpr int kSlabCells = kLayers * kG * kG;  // 1024 cells
constexpr int kBuildThreads = 1024;

__global__ void __launch_bounds__(kBuildThreads)
grid_build_kernel(int n, float inv_side, const float *__restrict__ xyz, int *__restrict__ cnt,
                  int *__restrict__ flags, float4 *__restrict__ slots,
                  float4 *__restrict__ ovf) {
  __shared__ int lcnt[kSlabCells];
  __shared__ int l_ovf;
  const BlockId blk = xcd_block_id();  // all 32 slabs of a cloud on one XCD: one HBM read
  const int slab = blk.x, b = blk.y;
  for (int t = threadIdx.x; t < kSlabCells; t += kBuildThreads) lcnt[t] = 0;
  if (threadIdx.x == 0) l_ovf = 0;
  __syncthreads();
  float4 *my_ovf = ovf + ((size_t)b * kSlabs + slab) * kOvfCap;
  const float *pts = xyz + (size_t)b * n * 3;
  const size_t cell0 = (size_t)b * kCellsPerCloud + (size_t)slab * kSlabCells;
  bool overflow = false;
  auto place = [&](float x, float y, float z, int k) {
    const int cz = cell_coord(z, inv_side) & (kG - 1);
    if (cz / kLayers != slab) return;
    const int local = ((cz % kLayers) * kG + (cell_coord(y, inv_side) & (kG - 1))) * kG +
                      (cell_coord(x, inv_side) & (kG - 1));
    const int slot = atomicAdd(&lcnt[local], 1);
    const float4 rec = make_float4(x, y, z, __builtin_bit_cast(float, k));
    if (slot < kCap) {
      slots[(cell0 + local) * kCap + slot] = rec;
    } else {  // dense cell: the point goes to this z-layer's overflow list
      const int o = atomicAdd(&l_ovf, 1);
      if (o < kOvfCap) my_ovf[o] = rec; else overflow = true;
    }
  };
  if ((n & 3) == 0) {
    // 4 points = 48 contiguous bytes = three 16-byte loads per lane, fully coalesced and with
    // no dependent second load; two groups in flight per lane
    const float4 *v = reinterpret_cast<const float4 *>(pts);
    const int groups = n / 4;
    for (int g0 = threadIdx.x; g0 < groups; g0 += 2 * kBuildThreads) {
      float4 a[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int g = g0 + u * kBuildThreads;
        if (g < groups) { a[u][0] = v[g * 3]; a[u][1] = v[g * 3 + 1]; a[u][2] = v[g * 3 + 2]; }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int g = g0 + u * kBuildThreads;
        if (g < groups) {
          place(a[u][0].x, a[u][0].y, a[u][0].z, g * 4 + 0);
          place(a[u][0].w, a[u][1].x, a[u][1].y, g * 4 + 1);
          place(a[u][1].z, a[u][1].w, a[u][2].x, g * 4 + 2);
          place(a[u][2].y, a[u][2].z, a[u][2].w, g * 4 + 3);
        }
      }
    }
  } else {
    for (int k = threadIdx.x; k < n; k += kBuildThreads)
      place(pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2], k);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < kSlabCells; t += kBuildThreads) cnt[cell0 + t] = lcnt[t];
  // flag + overflow length per (cloud, slab), always written: no clearing pass is needed
  const int any = __syncthreads_or(overflow ? 1 : 0);
  if (threadIdx.x == 0) {
    flags[(b * kSlabs + slab) * 2] = any;
    flags[(b * kSlabs + slab) * 2 + 1] = l_ovf < kOvfCap ? l_ovf : kOvfCap;
  }
}

// ---- 64-lane bitonic network on unsigned keys, entirely in registers ---------------------
template <int CTRL>
__device__ __forceinline__ unsigned dppu(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

// value held by lane (lane ^ D)
template <int D>
__device__ __forceinline__ unsigned partner(unsigned v, int lane) {
  if (D == 1) return dppu<0xB1>(v);                  // quad_perm [1,0,3,2]
  if (D == 2) return dppu<0x4E>(v);                  // quad_perm [2,3,0,1]
  if (D == 4) {                                      // row_shl:4 / row_shr:4
    const unsigned up = dppu<0x104>(v), dn = dppu<0x114>(v);
    return (lane & 4) ? dn : up;
  }
  if (D == 8) return dppu<0x128>(v);                 // row_ror:8
  unsigned r0, r1;
  if (D == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    r0 = r[0]; r1 = r[1];
    return (lane & 16) ? r0 : r1;
  }
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  r0 = r[0]; r1 = r[1];
  return (lane & 32) ? r0 : r1;
}

template <int D>
__device__ __forceinline__ unsigned cmpx(unsigned v, int lane, bool ascending) {
  const unsigned p = partner<D>(v, lane);
  const bool lower = (lane & D) == 0;
  const unsigned lo = v < p ? v : p, hi = v < p ? p : v;
  return (lower == ascending) ? lo : hi;
}

// merge a bitonic sequence of 64 keys into ascending (or descending) order
__device__ __forceinline__ unsigned bitonic_merge64(unsigned v, int lane, bool asc) {
  v = cmpx<32>(v, lane, asc); v = cmpx<16>(v, lane, asc); v = cmpx<8>(v, lane, asc);
  v = cmpx<4>(v, lane, asc);  v = cmpx<2>(v, lane, asc);  v = cmpx<1>(v, lane, asc);
  return v;
}

__device__ __forceinline__ unsigned bitonic_sort64(unsigned v, int lane, bool asc) {
#define DIR(K) (((lane & (K)) == 0) == asc)
  v = cmpx<1>(v, lane, DIR(2));
  v = cmpx<2>(v, lane, DIR(4)); v = cmpx<1>(v, lane, DIR(4));
  v = cmpx<4>(v, lane, DIR(8)); v = cmpx<2>(v, lane, DIR(8)); v = cmpx<1>(v, lane, DIR(8));
  v = cmpx<8>(v, lane, DIR(16)); v = cmpx<4>(v, lane, DIR(16)); v = cmpx<2>(v, lane, DIR(16));
  v = cmpx<1>(v, lane, DIR(16));
  v = cmpx<16>(v, lane, DIR(32)); v = cmpx<8>(v, lane, DIR(32)); v = cmpx<4>(v, lane, DIR(32));
  v = cmpx<2>(v, lane, DIR(32)); v = cmpx<1>(v, lane, DIR(32));
#undef DIR
  return bitonic_merge64(v, lane, asc);
}

// WIDE: nsample in (64, 128] -- two result registers per lane (the 64 smallest and the next 64)
template <bool WIDE>
__global__ void __launch_bounds__(256)
grid_query_kernel(int n, int m, float radius2, float inv_side, int nsample,
                  const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                  const int *__restrict__ cnt, const int *__restrict__ flags,
                  const float4 *__restrict__ slots, const float4 *__restrict__ ovf,
                  int *__restrict__ idx) {
  __shared__ unsigned hits[256 / kWave][kMaxHits];
  const BlockId blk = xcd_block_id();
  const int b = blk.y;
  const int lane = lane_id();
  const int wave = threadIdx.x / kWave;
  const int j = blk.x * (256 / kWave) + wave;
  if (j >= m) return;  // whole wave
  const float *pts = xyz + (size_t)b * n * 3;
  const float *ctr = new_xyz + ((size_t)b * m + j) * 3;
  int *row = idx + ((size_t)b * m + j) * nsample;
  const bool flagged = __ballot(lane < kSlabs && flags[(b * kSlabs + lane) * 2] != 0) != 0ull;
  if (flagged) {  // an overflow list of this cloud overflowed: exact brute-force scan instead
    ball_query_wave_scan<1>(pts, n, ctr, 1, radius2, nsample, row);
    return;
  }
  const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
  const int gx = cell_coord(cx, inv_side), gy = cell_coord(cy, inv_side),
            gz = cell_coord(cz, inv_side);

  // counts of the 27 neighbour cells: lane l <-> (dz, dy, dx) = (l/9, (l/3)%3, l%3) - 1
  int my_cell = 0, my_cnt = 0;
  if (lane < 27) {
    my_cell = cell_index(gx + lane % 3 - 1, gy + (lane / 3) % 3 - 1, gz + lane / 9 - 1);
    my_cnt = cnt[(size_t)b * kCellsPerCloud + my_cell];
    my_cnt = my_cnt < kCap ? my_cnt : kCap;
  }
  const float4 *cloud_slots = slots + (size_t)b * kCellsPerCloud * kCap;

  // ---- stream the nine x-rows; hits go to the LDS list in arrival order --------------------
  // Software-pipelined: the loads of row r+1 are issued before row r is tested, so a query
  // pays about five L2 round trips instead of nine.
  int total = 0;
  unsigned *list = hits[wave];
  struct Row { float4 q[kRowPasses]; int rc; };
  auto load_row = [&](int r, Row &o) {
    const int c0 = __builtin_amdgcn_readlane(my_cnt, r * 3 + 0);
    const int c1 = __builtin_amdgcn_readlane(my_cnt, r * 3 + 1);
    const int c2 = __builtin_amdgcn_readlane(my_cnt, r * 3 + 2);
    const int e0 = __builtin_amdgcn_readlane(my_cell, r * 3 + 0);
    const int e1 = __builtin_amdgcn_readlane(my_cell, r * 3 + 1);
    const int e2 = __builtin_amdgcn_readlane(my_cell, r * 3 + 2);
    const int c01 = c0 + c1;
    o.rc = c01 + c2;
#pragma unroll
    for (int p = 0; p < kRowPasses; ++p) {
      if (p * kWave < o.rc) {  // wave-uniform: the second / third pass of a row is rare
        // lanes past the end re-read the row's last candidate (a valid slot: no masked load, no
        // zero fill); they are excluded by `live` when the row is tested
        int t = p * kWave + lane;
        t = t < o.rc ? t : o.rc - 1;
        int cell = e0, s = t;
        if (t >= c0) { cell = e1; s = t - c0; }
        if (t >= c01) { cell = e2; s = t - c01; }
        o.q[p] = cloud_slots[cell * kCap + s];
      }
    }
  };
  Row rows[2];
  load_row(0, rows[0]);
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    if (r + 1 < 9) load_row(r + 1, rows[(r + 1) & 1]);
    const Row &cur = rows[r & 1];
#pragma unroll
    for (int p = 0; p < kRowPasses; ++p) {
      if (p * kWave < cur.rc) {  // wave-uniform
        const bool live = p * kWave + lane < cur.rc;
        const float d2 = sqdist3(cx, cy, cz, cur.q[p].x, cur.q[p].y, cur.q[p].z);
        const bool hit = live && d2 < radius2;
        const unsigned long long mask = __ballot(hit);
        if (mask) {
          const int pos = total + mask_rank(mask);
          if (hit && pos < kMaxHits) list[pos] = __builtin_bit_cast(unsigned, cur.q[p].w);
          total += __popcll(mask);
        }
      }
    }
  }
  // points that did not fit their cell sit in the overflow list of their z-layer: scan the
  // lists of the three layers around the centroid (empty unless the cloud has dense clumps)
#pragma unroll 1
  for (int dz = -1; dz <= 1; ++dz) {
    const int layer = (gz + dz) & (kG - 1);
    const int no = flags[(b * kSlabs + layer) * 2 + 1];
    const float4 *src = ovf + ((size_t)b * kSlabs + layer) * kOvfCap;
    for (int base = 0; base < no && total <= kMaxHits; base += kWave) {
      const bool live = base + lane < no;
      const float4 q = live ? src[base + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      const bool hit = live && sqdist3(cx, cy, cz, q.x, q.y, q.z) < radius2;
      const unsigned long long mask = __ballot(hit);
      if (mask) {
        const int pos = total + mask_rank(mask);
        if (hit && pos < kMaxHits) list[pos] = __builtin_bit_cast(unsigned, q.w);
        total += __popcll(mask);
      }
    }
  }
  if (total > kMaxHits) {  // very dense ball: exact brute-force scan for this centroid
    ball_query_wave_scan<1>(pts, n, ctr, 1, radius2, nsample, row);
    return;
  }
  if (total == 0) {
    for (int s = lane; s < nsample; s += kWave) row[s] = 0;
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- the 64 smallest indices, ascending, one per lane --------------------------------------
  const unsigned kNone = 0xFFFFFFFFu;
  unsigned a = lane < total ? list[lane] : kNone;
  a = bitonic_sort64(a, lane, true);
  for (int base = kWave; base < total; base += kWave) {
    unsigned c = base + lane < total ? list[base + lane] : kNone;
    c = bitonic_sort64(c, lane, false);   // descending: lane i holds the (63-i)-th smallest
    a = a < c ? a : c;                    // the 64 smallest of the union (a bitonic sequence)
    a = bitonic_merge64(a, lane, true);
  }
  const unsigned first = (unsigned)__builtin_amdgcn_readlane((int)a, 0);
  if (!WIDE) {
    const int have = total < kWave ? total : kWave;
    if (lane < nsample) row[lane] = (int)(lane < have ? a : first);
    return;
  }
  // WIDE: `a` holds the 64 smallest; a second pass over the list collects the next 64 -- the 64
  // smallest among the hits LARGER than a's maximum (indices are distinct, so "larger than the
  // 64th smallest" is exactly "not among the first 64")
  const unsigned cut = (unsigned)__builtin_amdgcn_readlane((int)a, 63);
  unsigned a1 = kNone;
  if (total > kWave) {
    bool started = false;
    for (int base = 0; base < total; base += kWave) {
      unsigned c = base + lane < total ? list[base + lane] : kNone;
      if (c <= cut) c = kNone;
      if (!started) {
        a1 = bitonic_sort64(c, lane, true);
        started = true;
      } else {
        c = bitonic_sort64(c, lane, false);
        a1 = a1 < c ? a1 : c;
        a1 = bitonic_merge64(a1, lane, true);
      }
    }
  }
  const int have = total < 2 * kWave ? total : 2 * kWave;
  if (lane < nsample) row[lane] = (int)(lane < have ? a : first);
  if (kWave + lane < nsample) row[kWave + lane] = (int)(kWave + lane < have ? a1 : first);
}

}  // namespace

size_t pn2_ball_query_grid_workspace(int b, int n, int m, int nsample) {
  (void)m;
  if (n < 4096 || nsample > 2 * kWave) return 0;
  return grid_cnt_bytes(b) + sizeof(float4) * (size_t)b * kCellsPerCloud * kCap +
         sizeof(float4) * (size_t)b * kSlabs * kOvfCap;
}

int pn2_ball_query_grid_try(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                            const float *xyz, int *idx, void *workspace, size_t workspace_bytes,
                            hipStream_t stream, int *handled) {
  *handled = 0;
  const size_t need = pn2_ball_query_grid_workspace(b, n, m, nsample);
  if (need == 0 || workspace == nullptr || workspace_bytes < need) return 0;
  if (!(radius > 1e-6f) || !(radius < 1e6f)) return 0;  // also rejects NaN
  char *ws = reinterpret_cast<char *>(workspace);
  int *cnt = reinterpret_cast<int *>(ws);
  int *flags = cnt + (size_t)b * kCellsPerCloud;
  float4 *slots = reinterpret_cast<float4 *>(ws + grid_cnt_bytes(b));
  float4 *ovf = slots + (size_t)b * kCellsPerCloud * kCap;
  const float inv_side = 1.0f / (radius * 1.001f);
  hipLaunchKernelGGL(grid_build_kernel, dim3(kSlabs, b), dim3(kBuildThreads), 0, stream, n,
                     inv_side, xyz, cnt, flags, slots, ovf);
  const float radius2 = radius * radius;  // fp32 product, as ball_query_gpu.cu:27
  if (nsample > kWave)
    hipLaunchKernelGGL(grid_query_kernel<true>, dim3(pn2_ceil_div(m, 256 / kWave), b), dim3(256),
                       0, stream, n, m, radius2, inv_side, nsample, new_xyz, xyz, cnt, flags,
                       slots, ovf, idx);
  else
    hipLaunchKernelGGL(grid_query_kernel<false>, dim3(pn2_ceil_div(m, 256 / kWave), b), dim3(256),
                       0, stream, n, m, radius2, inv_side, nsample, new_xyz, xyz, cnt, flags,
                       slots, ovf, idx);
  *handled = 1;
  return pn2_launch_status();
}
