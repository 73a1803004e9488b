// 3dioumatch_amd/csrc/pn2_ball_tile.hip -- ball_query (+ the gathers of QueryAndGroup) on the
// cell lists of pn2_ball_grid.hip, TILE form: the candidates of neighbouring centroids are staged
// in LDS once.
//
// Semantics: ball_query_gpu.cu:14-49 (first nsample indices in ascending order with d2 < r^2,
// tail padded with the first hit, zero row without a hit); group_points_gpu.cu:13-33 and
// pointnet2_utils.py:348-358 for the fused gather; SURVEY App. A.3/A.4.
//
// Why: the wave-per-centroid kernels fetch nine 1 KB rows per centroid through the vector memory
// path; measured (profiles/r3_pair_mem_pipeline_counters.csv) that path -- the per-CU address /
// data-return units, TD busy 73 % of the kernel -- and not instruction issue bounds them: doubling
// the resident waves doubled every stage's latency and left the kernel where it was.  Centroids of
// the same 2x2x2 block of cells (a TILE, ~6 of them at the SA1 density) share the same 4x4x4
// cells of candidates (the HALO, ~1000 records), so
//   plan  : the centroids are counting-sorted by tile once (grid_plan_kernel -- or the tail of the
//           furthest-point-sampling kernel of the same layer, which picked them): `order`, and per
//           occupied tile (tile id, first position, count);
//   query : a workgroup of eight waves takes a tile, copies the 64 halo cells from the CSR records
//           into LDS with coalesced loads (each record crosses the vector memory path once per
//           tile instead of ~2.4 times per centroid), then every wave answers one centroid at a
//           time against LDS: nine broadcast reads give its row ranges, nine ds_read_b128 its
//           candidates; hits are kept as 32-bit record positions, ranked by index with the
//           bucket ranking of pn2_ball_grid.hip, and the selected records are read back from the
//           halo for the output row.
// Anything unusual about a ball (a row of more than 64 candidates, more hits than the list holds,
// no hit, a halo that does not fit, a centroid that is not in the tile the plan says) takes a
// general path over the global records with the same list / ranking code.
#include "common.h"
#include "ball_common.h"
#include "grid_common.h"
#include "tile_plan.h"

namespace {

using namespace grid;

constexpr int kTileWaves = 8;                   // waves per workgroup of the query kernel
constexpr int kHaloCap = 1152;                  // records of a halo kept in LDS (18 KB: four
                                                // workgroups of 38 KB per CU with room to spare)
constexpr int kMaxHits = 192;                   // capacity of a wave's hit list (nsample <= 64)
__global__ void __launch_bounds__(kPlanThreads)
grid_plan_kernel(int m, float inv_side, const float *__restrict__ new_xyz, int *__restrict__ plan) {
  __shared__ int hist[kTiles];
  __shared__ int scr[32];
  const float *c = new_xyz + (size_t)blockIdx.x * m * 3;
  plan_build(m, inv_side,
             [&](int j, float &x, float &y, float &z) { x = c[j * 3]; y = c[j * 3 + 1]; z = c[j * 3 + 2]; },
             hist, scr, plan + (size_t)blockIdx.x * (4 + 3 * (size_t)m));
}

// ---- the query ----------------------------------------------------------------------------------
struct TileGroupOut {     // fused gather
  const float *features;  // (b, c, n) or nullptr
  float *out;             // (b, ctot, m, ns)
  int c, ctot, normalize;
  float inv_radius;
};

struct alignas(16) TileWaveLds {
  unsigned list[kMaxHits + 1];   // record positions of the hits, arrival order; [kMaxHits]: dump
  unsigned tmp[kMaxHits + 16];   // keys grouped by bucket, then sentinels
  unsigned char perm[kMaxHits];  // perm[rank] = list position
  int cnt[kWave];
  int off[kWave];
};

struct alignas(16) TileLds {
  float4 halo[kHaloCap + kWave];  // + 64: whole-row reads past a row's end stay inside
  int hoff[68];                   // first halo position of halo cell l (hoff[64] = total)
  TileWaveLds wave[kTileWaves];
};

template <class T>
__device__ __forceinline__ void put_t(T *p, T v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// rank the `total` listed hits of this wave by index (see rank_hits in pn2_ball_grid.hip);
// key_of(list position) -> index of that hit
template <class KeyOf>
__device__ __forceinline__ void tile_rank(TileWaveLds &L, int total, int have, unsigned bucket_mul,
                                          int lane, KeyOf key_of) {
  constexpr int TMAX = kMaxHits / kWave;
  L.cnt[lane] = 0;
  if (lane < 16) L.tmp[total + lane] = 0xffffffffu;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  unsigned key[TMAX];
  int bk[TMAX], slot[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    bk[t] = -1;
    if (t * kWave < total) {
      const int e = t * kWave + lane;
      if (e < total) {
        key[t] = key_of(e);
        bk[t] = (int)__umulhi(key[t], bucket_mul);
        slot[t] = atomicAdd(&L.cnt[bk[t]], 1);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  bool big;
  {
    const int cn = L.cnt[lane];
    L.off[lane] = wave_scan_incl(cn) - cn;
    big = __builtin_amdgcn_ballot_w64(cn > 4) != 0ull;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int first[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t)
    if (t * kWave < total && bk[t] >= 0) {
      first[t] = L.off[bk[t]];
      L.tmp[first[t] + slot[t]] = key[t];
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int t = 0; t < TMAX; ++t)
    if (t * kWave < total && bk[t] >= 0) {
      const unsigned *w = &L.tmp[first[t]];
      int rank = first[t];
      rank += w[0] < key[t] ? 1 : 0;
      rank += w[1] < key[t] ? 1 : 0;
      rank += w[2] < key[t] ? 1 : 0;
      rank += w[3] < key[t] ? 1 : 0;
      if (big) {
        const int sz = L.cnt[bk[t]];
#pragma clang loop vectorize(disable) unroll(disable)
        for (int u = 4; u < sz; ++u) rank += w[u] < key[t] ? 1 : 0;
      }
      if (rank < have) L.perm[rank] = (unsigned char)(t * kWave + lane);
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <bool GROUP, bool PROF>
__global__ void __launch_bounds__(kTileWaves *kWave) __attribute__((amdgpu_waves_per_eu(8, 8)))
grid_tile_query_kernel(int n, int m, int wg_per_cloud, float radius2, float inv_side, int nsample,
                       unsigned bucket_mul, int flags, const float *__restrict__ new_xyz,
                       const float *__restrict__ xyz, const int *__restrict__ start,
                       const float4 *__restrict__ rec, const int *__restrict__ plan_all,
                       int *__restrict__ idx, TileGroupOut g,
                       unsigned long long *__restrict__ prof) {
  __shared__ TileLds S;
  // stage clocks of this wave (tools/pair_bench.py --stages)
  unsigned long long t_start = 0, t_prev = 0, acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned n_tiles = 0, n_cen = 0, n_general = 0;  // (PROF) work of this wave
  auto stamp = [&](int q) {
    if (PROF) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      acc[q] += now - t_prev;
      t_prev = now;
    }
  };
  if (PROF) t_start = t_prev = __builtin_amdgcn_s_memtime();
  const int wg = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x);
  const int b = wg / wg_per_cloud, g_in_cloud = wg - b * wg_per_cloud;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const bool nt = (flags & 2) != 0;
  TileWaveLds &L = S.wave[wave];
  const float *pts = xyz + (size_t)b * n * 3;
  const int *st = start + (size_t)b * kStartStride;
  const float4 *cloud = rec + (size_t)b * n;
  const int *plan = plan_all + (size_t)b * (4 + 3 * (size_t)m);
  const int *order = plan + 4, *trange = plan + 4 + m;
  const size_t plane = (size_t)m * nsample;
  const int ntiles = plan[0];

  // the record of a tile and the CSR range of this lane's halo cell travel one tile ahead
  auto tile_cell_range = [&](int tw_, int &hs_, int &hcnt_) {
    const int tile_ = tw_ & (kTiles - 1);
    const int ox_ = (tile_ & (kTilesPerAxis - 1)) * 2, oy_ = ((tile_ / kTilesPerAxis) & (kTilesPerAxis - 1)) * 2;
    const int oz_ = (tile_ / (kTilesPerAxis * kTilesPerAxis)) * 2;
    const int hx = lane & 3, hy = (lane >> 2) & 3, hz = lane >> 4;
    const int cell = (((oz_ - 1 + hz) & (kG - 1)) * kG + ((oy_ - 1 + hy) & (kG - 1))) * kG +
                     ((ox_ - 1 + hx) & (kG - 1));
    hs_ = st[cell];
    hcnt_ = st[cell + 1] - hs_;
  };
  int k = g_in_cloud;
  int tw = 0, cbegin = 0, hs = 0, hcnt = 0;
  if (k < ntiles) {
    tw = trange[2 * k];
    cbegin = trange[2 * k + 1];
    tile_cell_range(tw, hs, hcnt);
  }
#pragma unroll 1
  while (k < ntiles) {
    const int tile = tw & (kTiles - 1), ccount = tw >> 12;
    const int ox = (tile & (kTilesPerAxis - 1)) * 2, oy = ((tile / kTilesPerAxis) & (kTilesPerAxis - 1)) * 2;
    const int oz = (tile / (kTilesPerAxis * kTilesPerAxis)) * 2;
    // this wave's first centroid: its coordinates travel while the halo is staged
    int jn = -1;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (wave < ccount) {
      jn = order[cbegin + wave];
      const float *cp = new_xyz + ((size_t)b * m + jn) * 3;
      nx = cp[0]; ny = cp[1]; nz = cp[2];
    }
    // the next tile's record
    const int k2 = k + wg_per_cloud;
    int tw2 = 0, cbegin2 = 0, hs2 = 0, hcnt2 = 0;
    if (k2 < ntiles) {
      tw2 = trange[2 * k2];
      cbegin2 = trange[2 * k2 + 1];
    }
    // ---- halo: 64 cells, lane = cell (every wave computes the offsets, wave 0 publishes) -------
    int hofs, htotal;
    {
      const int incl = wave_scan_incl(hcnt);
      hofs = incl - hcnt;
      htotal = __builtin_amdgcn_readlane(incl, kWave - 1);
      if (wave == 0) {
        S.hoff[lane] = hofs;
        if (lane == kWave - 1) S.hoff[kWave] = incl;
      }
    }
    const bool fits = htotal <= kHaloCap;  // wave-uniform, same in every wave
    stamp(0);  // tile record, cell offsets, scan
    if (fits) {
      // wave w copies cells w*8 .. w*8+7, eight lanes per cell
      const int cl = wave * 8 + (lane >> 3), sub = lane & 7;
      const int cs = __shfl(hs, cl, kWave), cc = __shfl(hcnt, cl, kWave), co = __shfl(hofs, cl, kWave);
      for (int u = sub; u < cc; u += 8) S.halo[co + u] = cloud[cs + u];
    }
    __syncthreads();
    stamp(1);  // halo copy + barrier
    if (PROF) ++n_tiles;
    if (k2 < ntiles) tile_cell_range(tw2, hs2, hcnt2);  // lands while the centroids are answered

    // ---- centroids of the tile: wave w takes w, w + 8, ... ------------------------------------
#pragma unroll 1
    for (int i = wave; i < ccount; i += kTileWaves) {
      const int j = jn;
      const float cx = nx, cy = ny, cz = nz;
      if (i + kTileWaves < ccount) {  // the next one of this wave
        jn = order[cbegin + i + kTileWaves];
        const float *cp = new_xyz + ((size_t)b * m + jn) * 3;
        nx = cp[0]; ny = cp[1]; nz = cp[2];
      }
      int *row = idx + ((size_t)b * m + j) * nsample;
      const int gx = __builtin_amdgcn_readfirstlane(cell_coord(cx, inv_side)) & (kG - 1);
      const int gy = __builtin_amdgcn_readfirstlane(cell_coord(cy, inv_side)) & (kG - 1);
      const int gz = __builtin_amdgcn_readfirstlane(cell_coord(cz, inv_side)) & (kG - 1);
      const int lx = gx - ox, ly = gy - oy, lz = gz - oz;
      bool general = !fits || ((lx | ly | lz) & ~1) != 0;  // halo too large / not this tile's
      int total = 0;
      if (!general) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 cxy = {cx, cy};
        // all eighteen range bounds first (one LDS round trip), then the candidates three rows at
        // a time: a row-by-row loop pays two dependent LDS round trips per row
        int rs[9], rl[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          const int c0 = ((lz + r / 3) * 4 + (ly + r % 3)) * 4 + lx;  // halo cell of the row's start
          rs[r] = S.hoff[c0];
          rl[r] = S.hoff[c0 + 3];
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          rl[r] -= rs[r];
          general |= rl[r] > kWave;
          rs[r] += lane;
        }
#pragma unroll
        for (int r3 = 0; r3 < 9; r3 += 3) {
          float4 q[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) q[u] = S.halo[rs[r3 + u]];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int r = r3 + u;
            const f2 qxy = {q[u].x, q[u].y};
            const f2 d = cxy - qxy;
            const f2 dd = d * d;
            const float dz = __fsub_rn(cz, q[u].z);
            const float d2 = __fadd_rn(__fadd_rn(dd.x, dd.y), __fmul_rn(dz, dz));
            const bool near = d2 < radius2, inrow = lane < rl[r];
            const unsigned long long mask =
                __builtin_amdgcn_ballot_w64(near) & __builtin_amdgcn_ballot_w64(inrow);
            const int at = total + mask_rank(mask);
            if (near & inrow) L.list[at < kMaxHits ? at : kMaxHits] = (unsigned)rs[r];
            total += __popcll(mask);
          }
        }
        general = __builtin_amdgcn_readfirstlane((int)general) != 0;
        general = general || total > kMaxHits || total == 0;
      }
      stamp(2);  // row ranges, candidates, tests, list
      if (PROF) { ++n_cen; n_general += general ? 1u : 0u; }
      bool from_halo = true;
      bool done = false;
      if (general) {
        // ---- general path: every row range of the global CSR, any length, the wrapped cell at
        //      the lattice seam; list entries are positions in the cloud's record array
        from_halo = false;
        total = 0;
        const int xa = gx > 0 ? gx - 1 : 0, xb = gx < kG - 1 ? gx + 1 : kG - 1;
        auto scan_range = [&](int from, int to) {
          for (int p0 = from; p0 < to; p0 += kWave) {
            const int p = p0 + lane;
            float4 qq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < to) qq = cloud[p];
            const bool h = p < to && sqdist3(cx, cy, cz, qq.x, qq.y, qq.z) < radius2;
            const unsigned long long hm = __builtin_amdgcn_ballot_w64(h);
            const int pos = total + mask_rank(hm);
            if (h && pos < kMaxHits) L.list[pos] = (unsigned)p;
            total += __popcll(hm);
          }
        };
#pragma unroll 1
        for (int r = 0; r < 9; ++r) {
          const int rowbase = (((gz + r / 3 - 1) & (kG - 1)) * kG + ((gy + r % 3 - 1) & (kG - 1))) * kG;
          scan_range(st[rowbase + xa], st[rowbase + xb + 1]);
          if (gx == 0 || gx == kG - 1) {
            const int wc = rowbase + (gx == 0 ? kG - 1 : 0);
            scan_range(st[wc], st[wc + 1]);
          }
        }
        if (total > kMaxHits) {  // very dense ball: exact brute-force scan
          const float *ctr = new_xyz + ((size_t)b * m + j) * 3;
          ball_query_wave_scan<1>(pts, n, ctr, 1, radius2, nsample, row);
          if (GROUP) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // this wave's stores -> its loads
            if (lane < nsample) {
              const int v = row[lane];
              const float4 rr = make_float4(pts[v * 3 + 0], pts[v * 3 + 1], pts[v * 3 + 2],
                                            __builtin_bit_cast(float, v));
              float *ob = g.out + (size_t)b * g.ctot * plane + (size_t)j * nsample;
              float rx = __fsub_rn(rr.x, cx), ry = __fsub_rn(rr.y, cy), rz = __fsub_rn(rr.z, cz);
              if (g.normalize) {
                rx = __fmul_rn(rx, g.inv_radius); ry = __fmul_rn(ry, g.inv_radius); rz = __fmul_rn(rz, g.inv_radius);
              }
              ob[lane] = rx; ob[plane + lane] = ry; ob[2 * plane + lane] = rz;
              for (int l = 0; l < g.c; ++l)
                ob[(size_t)(3 + l) * plane + lane] = g.features[((size_t)b * g.c + l) * n + v];
            }
          }
          done = true;
        } else if (total == 0) {  // no hit: the reference's zero-initialised row -> point 0
          if (lane < nsample) {
            row[lane] = 0;
            if (GROUP) {
              float *ob = g.out + (size_t)b * g.ctot * plane + (size_t)j * nsample;
              float rx = __fsub_rn(pts[0], cx), ry = __fsub_rn(pts[1], cy), rz = __fsub_rn(pts[2], cz);
              if (g.normalize) {
                rx = __fmul_rn(rx, g.inv_radius); ry = __fmul_rn(ry, g.inv_radius); rz = __fmul_rn(rz, g.inv_radius);
              }
              ob[lane] = rx; ob[plane + lane] = ry; ob[2 * plane + lane] = rz;
              for (int l = 0; l < g.c; ++l)
                ob[(size_t)(3 + l) * plane + lane] = g.features[((size_t)b * g.c + l) * n];
            }
          }
          done = true;
        }
      }
      if (!done) {
        const int have = total < nsample ? total : nsample;
        if (from_halo)
          tile_rank(L, total, have, bucket_mul, lane,
                    [&](int e) { return __builtin_bit_cast(unsigned, S.halo[L.list[e]].w); });
        else
          tile_rank(L, total, have, bucket_mul, lane,
                    [&](int e) { return __builtin_bit_cast(unsigned, cloud[L.list[e]].w); });
        stamp(3);  // ranking
        const unsigned p = L.list[L.perm[lane < have ? lane : 0]];  // tail: first hit
        const float4 rr = from_halo ? S.halo[p] : cloud[p];
        if (lane < nsample) {
          const unsigned v = __builtin_bit_cast(unsigned, rr.w);
          put_t(row + lane, (int)v, nt);
          if (GROUP) {
            float *ob = g.out + (size_t)b * g.ctot * plane + (size_t)j * nsample;
            float rx = __fsub_rn(rr.x, cx), ry = __fsub_rn(rr.y, cy), rz = __fsub_rn(rr.z, cz);
            if (g.normalize) {
              rx = __fmul_rn(rx, g.inv_radius); ry = __fmul_rn(ry, g.inv_radius); rz = __fmul_rn(rz, g.inv_radius);
            }
            put_t(ob + lane, rx, nt);
            put_t(ob + plane + lane, ry, nt);
            put_t(ob + 2 * plane + lane, rz, nt);
            for (int l = 0; l < g.c; ++l)
              put_t(ob + (size_t)(3 + l) * plane + lane, g.features[((size_t)b * g.c + l) * n + v], nt);
          }
        }
      }
      // the next centroid reuses this wave's LDS
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      stamp(4);  // slot reads, gather, stores issued (+ the general path)
    }
    __syncthreads();  // the halo is replaced by the next tile's
    stamp(5);  // waiting for the tile's other waves
    k = k2; tw = tw2; cbegin = cbegin2; hs = hs2; hcnt = hcnt2;
  }
  if (PROF && prof != nullptr && lane == 0) {
    unsigned long long *o = prof + ((size_t)wg * kTileWaves + wave) * 8;
    for (int q = 0; q < 6; ++q) o[q] = acc[q];
    o[6] = t_prev - t_start;
    o[7] = (unsigned long long)n_tiles | ((unsigned long long)n_cen << 16) |
           ((unsigned long long)n_general << 32);
  }
}

}  // namespace

// ints of a plan for b clouds of m centroids
size_t pn2_tile_plan_ints(int b, int m) { return (size_t)b * (4 + 3 * (size_t)m); }

int pn2_tile_plan_supported(int n, int m, int nsample) {
  return n >= 4096 && n <= kGridMaxPoints && m >= 1 && m <= 65535 && nsample >= 1 && nsample <= kWave;
}

int pn2_tile_plan_launch(int b, int m, float radius, const float *new_xyz, int *plan,
                         hipStream_t stream) {
  hipLaunchKernelGGL(grid_plan_kernel, dim3(b), dim3(kPlanThreads), 0, stream, m,
                     grid_inv_side(radius), new_xyz, plan);
  return pn2_launch_status();
}

// the plan for the centroids xyz[idxs[j]] (used by the sampling kernel's tail: pn2_fps_bucket.hip
// includes this file's plan_build through pn2_tile_plan.h)
int pn2_tile_query_launch(int b, int n, int m, int c_gather, int ctot, float radius, int nsample,
                          int normalize_xyz, int flags, const float *new_xyz, const float *xyz,
                          const float *features, int *idx, float *out, void *grid_ws,
                          const int *plan, unsigned long long *prof, hipStream_t stream) {
  const GridWs ws = grid_ws_layout(grid_ws, b, n);
  const float radius2 = radius * radius;  // fp32 product, as ball_query_gpu.cu:27
  const unsigned bucket_mul = (unsigned)(((unsigned long long)64 << 32) / (unsigned long long)n);
  TileGroupOut g = {features, out, c_gather, ctot, normalize_xyz, 1.0f / radius};
  // four workgroups of eight waves per CU; a cloud's workgroups share an XCD (its L2 holds the
  // cloud's records)
  int wpc = (256 * 4) / (b > 0 ? b : 1);
  if (wpc < 1) wpc = 1;
  if (wpc > 512) wpc = 512;
#define TILE_QUERY(GROUP, PROF)                                                                    \
  hipLaunchKernelGGL((grid_tile_query_kernel<GROUP, PROF>), dim3(wpc * b),                         \
                     dim3(kTileWaves * kWave), 0, stream, n, m, wpc, radius2,                      \
                     grid_inv_side(radius), nsample, bucket_mul, flags, new_xyz, xyz, ws.start,    \
                     ws.rec, plan, idx, g, prof)
  if ((flags & 4) && prof && out) TILE_QUERY(true, true);
  else if (out) TILE_QUERY(true, false);
  else TILE_QUERY(false, false);
#undef TILE_QUERY
  return pn2_launch_status();
}
