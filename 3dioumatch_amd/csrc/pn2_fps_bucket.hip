// 3dioumatch_amd/csrc/pn2_fps_bucket.hip -- exact furthest point sampling for LARGE clouds
// (SA1: 40 000 -> 2048) in one workgroup per cloud, with spatial pruning.
//
// Why: FPS is m-1 strictly dependent rounds.  The reference (sampling_gpu.cu:75-178) and the
// streaming tier of pn2_sampling.hip re-read every point every round (N*16 bytes per round
// from L2: 8.9 us per round at N = 40 000, 18 ms per call -- 40 % of a train step).  But a new
// sample only lowers the running distance of points that are closer to it than their current
// distance, and those are spatially clustered.  So:
//
//   * once per call the cloud is counting-sorted by a coarse Morton cell (16^3 cells, LDS
//     histogram + LDS atomic cursors) into a scratch array of (x, y, z, original index), and
//     cut into BUCKETS of 64 consecutive sorted points.  Bucket b belongs to wave b % 16;
//     lane l of that wave holds point l of the bucket, and keeps ITS running distance in a
//     register for the whole call (NBW buckets per wave -> NBW registers per lane);
//   * lane j of a wave also holds the metadata of the wave's j-th bucket: tight bounding box,
//     current maximum running distance and the point that attains it (index + coordinates);
//   * a round = (1) every lane tests its bucket: if the squared distance from the new sample
//     to the bounding box (shrunk by 1e-6 relative, to stay conservative under fp32 rounding)
//     is not below the bucket's maximum, no point of the bucket can change -> skip;
//     (2) the wave visits only the ballot-selected buckets: one coalesced 1 KiB load, the
//     reference's exact fp32 update, a 64-lane argmax (DPP / permlane, fps_common.h);
//     (3) argmax over the per-bucket maxima of the wave, then across the 16 waves through LDS
//     with one barrier (fps_block_pick).
//   After a few dozen rounds only ~1 bucket per wave is touched per round.
//
// Exactness: skipping is a no-op by construction (min(d, t) == t for every point of a skipped
// bucket), the update arithmetic is the reference's, and ties are resolved by the reference's
// reduction-tree order through fps_key -- so the arrangement of points in buckets (which
// depends on atomic ordering) cannot influence the result.
#include "common.h"
#include "fps_common.h"
#include "grid_common.h"

namespace {

using namespace fps;

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / kWave;  // 16
constexpr int kGrid = 16;                 // cells per axis
constexpr int kCells = kGrid * kGrid * kGrid;

// (NOT -wave_max_f32(-v): clang 22 / ROCm 7.2 folds the negations into source modifiers of the
//  v_cndmask_b32 selects of the DPP steps and the result came out wrong for negative inputs --
//  bounding boxes of clouds with negative coordinates were too tight, buckets were pruned that
//  still held the farthest point: regression test test_fps_negative_coordinates)
__device__ __forceinline__ float wave_min_f32(float v) {
#define FPS_MIN_STEP(CTRL) { const float o = __builtin_bit_cast(float, dpp_mov<CTRL>(__builtin_bit_cast(unsigned, v))); v = o < v ? o : v; }
  FPS_MIN_STEP(kQuadXor1) FPS_MIN_STEP(kQuadXor2) FPS_MIN_STEP(kRowHalfMirror) FPS_MIN_STEP(kRowMirror)
#undef FPS_MIN_STEP
  unsigned r0, r1;
  swap_rows<false>(__builtin_bit_cast(unsigned, v), r0, r1);
  float f0 = __builtin_bit_cast(float, r0), f1 = __builtin_bit_cast(float, r1);
  v = f0 < f1 ? f0 : f1;
  swap_rows<true>(__builtin_bit_cast(unsigned, v), r0, r1);
  f0 = __builtin_bit_cast(float, r0); f1 = __builtin_bit_cast(float, r1);
  return f0 < f1 ? f0 : f1;
}

__device__ __forceinline__ unsigned spread4(unsigned v) {  // abcd -> 00a00b00c00d
  v &= 0xF;
  v = (v | (v << 4)) & 0x0C3;   // 0000 ab00 00cd  -> ..ab....cd
  v = (v | (v << 2)) & 0x249;   // a..b..c..d
  return v;
}

__device__ __forceinline__ int morton_cell(float x, float y, float z, const float *lo,
                                           const float *inv) {
  int cx = (int)((x - lo[0]) * inv[0]);
  int cy = (int)((y - lo[1]) * inv[1]);
  int cz = (int)((z - lo[2]) * inv[2]);
  cx = cx < 0 ? 0 : (cx > kGrid - 1 ? kGrid - 1 : cx);
  cy = cy < 0 ? 0 : (cy > kGrid - 1 ? kGrid - 1 : cy);
  cz = cz < 0 ? 0 : (cz > kGrid - 1 ? kGrid - 1 : cz);
  return (int)(spread4((unsigned)cx) | (spread4((unsigned)cy) << 1) | (spread4((unsigned)cz) << 2));
}

// squared distance from p to an axis-aligned box (0 inside)
__device__ __forceinline__ float box_dist2(float px, float py, float pz, float lx, float ly,
                                           float lz, float hx, float hy, float hz) {
  const float dx = fmaxf(fmaxf(lx - px, px - hx), 0.f);
  const float dy = fmaxf(fmaxf(ly - py, py - hy), 0.f);
  const float dz = fmaxf(fmaxf(lz - pz, pz - hz), 0.f);
  return dx * dx + dy * dy + dz * dz;
}

// My point of the wave's jj-th bucket.  The byte offset is rebuilt at every use from one
// lane offset and one compile-time constant hidden behind empty asm statements; otherwise the
// compiler hoists NBW loop-invariant 64-bit addresses out of the round loop and spends two
// VGPRs per bucket on them (spilling at NBW >= 40).
__device__ __forceinline__ float4 load_bucket(const float4 *wave_base, int jj, unsigned lane_off) {
  unsigned so = (unsigned)jj * (kWaves * kWave * (unsigned)sizeof(float4));
  asm volatile("" : "+s"(so));
  asm volatile("" : "+v"(lane_off));
  const char *p = reinterpret_cast<const char *>(wave_base) + so;
  return *reinterpret_cast<const float4 *>(p + lane_off);
}

template <int NBW>
__global__ void __launch_bounds__(kThreads)
fps_bucket_kernel(int n, int m, int log2bs, const float *__restrict__ dataset,
                  float4 *__restrict__ scratch, int *__restrict__ idxs, float grid_inv_side,
                  int *__restrict__ grid_start, float4 *__restrict__ grid_rec) {
  __shared__ int cell_cnt[kCells];                 // histogram, then scatter cursors
  __shared__ float red[kWaves * 8];
  __shared__ __attribute__((aligned(16))) float slots[2][kWaves * 8];
  __shared__ float box[8];                         // lo[3], inv[3]
  __shared__ int s_valid;

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int w = tid / kWave;
  const float *pts = dataset + (size_t)blockIdx.x * n * 3;
  float4 *sp = scratch + (size_t)blockIdx.x * (kThreads * NBW);
  int *out = idxs + (size_t)blockIdx.x * m;

  // ---- G: by-product -- the ball-query cell lists of this cloud (grid_common.h) -------------
  // The set-abstraction layer that samples this cloud queries balls in it right afterwards.
  // This workgroup streams the cloud anyway (and has ~6 ms of serial rounds ahead of it), so it
  // also counting-sorts the points by lattice cell: 32768 16-bit counters packed in 64 KB of LDS
  // (n < 65536 on this path), scan, scatter.  The separate two-kernel build of the ball query
  // (15 us on all CUs) disappears from the layer.
  if (grid_start != nullptr) {
    __shared__ unsigned gcnt[grid::kCells / 2];
    __shared__ int g_wave_tot[kWaves];
    for (int t = tid; t < grid::kCells / 2; t += kThreads) gcnt[t] = 0u;
    __syncthreads();
    for (int k = tid; k < n; k += kThreads) {
      const int cell = grid::cell_id(pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2], grid_inv_side);
      atomicAdd(&gcnt[cell >> 1], 1u << ((cell & 1) * 16));
    }
    __syncthreads();
    {  // exclusive scan over the 32768 cells: 32 consecutive cells (16 words) per lane
      unsigned wds[16];
      int sum = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        wds[q] = gcnt[tid * 16 + q];
        sum += (int)(wds[q] & 0xFFFFu) + (int)(wds[q] >> 16);
      }
      int incl = sum;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
      }
      if (lane == kWave - 1) g_wave_tot[w] = incl;
      __syncthreads();
      int run = incl - sum;
      for (int q = 0; q < w; ++q) run += g_wave_tot[q];
      int *st = grid_start + (size_t)blockIdx.x * grid::kStartStride + tid * 32;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int lo = run, hi = run + (int)(wds[q] & 0xFFFFu);
        st[2 * q] = lo;
        st[2 * q + 1] = hi;
        gcnt[tid * 16 + q] = (unsigned)lo | ((unsigned)hi << 16);  // scatter cursors (n < 65536)
        run = hi + (int)(wds[q] >> 16);
      }
      if (tid == kThreads - 1) grid_start[(size_t)blockIdx.x * grid::kStartStride + grid::kCells] = run;
    }
    __syncthreads();
    float4 *rec = grid_rec + (size_t)blockIdx.x * n;
    for (int k = tid; k < n; k += kThreads) {
      const float x = pts[k * 3 + 0], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
      const int cell = grid::cell_id(x, y, z, grid_inv_side);
      const int sh = (cell & 1) * 16;
      const unsigned old = atomicAdd(&gcnt[cell >> 1], 1u << sh);
      rec[(old >> sh) & 0xFFFFu] = make_float4(x, y, z, __builtin_bit_cast(float, k));
    }
    __syncthreads();
  }

  // ---- P0: bounding box of the participating points --------------------------------------
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int k = tid; k < n; k += kThreads) {
    const float x = pts[k * 3 + 0], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
    if (!fps_skipped(x, y, z)) {
      lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
      hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
    }
  }
  for (int t = tid; t < kCells; t += kThreads) cell_cnt[t] = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = wave_min_f32(lo[d]);
    hi[d] = wave_max_f32(hi[d]);
  }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { red[w * 8 + d] = lo[d]; red[w * 8 + 3 + d] = hi[d]; }
  }
  __syncthreads();
  if (tid < 3) {
    float l = red[tid], h = red[3 + tid];
    for (int q = 1; q < kWaves; ++q) {
      l = fminf(l, red[q * 8 + tid]);
      h = fmaxf(h, red[q * 8 + 3 + tid]);
    }
    const float ext = h - l;
    box[tid] = l;
    box[3 + tid] = ext > 0.f ? (float)kGrid / ext : 0.f;
  }
  __syncthreads();

  // ---- P1: histogram over Morton cells ----------------------------------------------------
  for (int k = tid; k < n; k += kThreads) {
    const float x = pts[k * 3 + 0], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
    if (!fps_skipped(x, y, z)) atomicAdd(&cell_cnt[morton_cell(x, y, z, box, box + 3)], 1);
  }
  __syncthreads();

  // ---- P2: exclusive scan of the 4096 counters (4 per thread) ----------------------------
  {
    int c[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { c[q] = cell_cnt[tid * 4 + q]; sum += c[q]; }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int o = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += o;
    }
    int *wave_tot = reinterpret_cast<int *>(red);
    if (lane == kWave - 1) wave_tot[w] = incl;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < w; ++q) base += wave_tot[q];
    if (tid == kThreads - 1) s_valid = base + incl;
    int run = base + incl - sum;
#pragma unroll
    for (int q = 0; q < 4; ++q) { cell_cnt[tid * 4 + q] = run; run += c[q]; }
  }
  __syncthreads();
  const int n_valid = s_valid;
  const int n_buckets = (n_valid + kWave - 1) / kWave;

  // ---- P3: scatter into sorted order; pad the last bucket ---------------------------------
  for (int k = tid; k < n; k += kThreads) {
    const float x = pts[k * 3 + 0], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
    if (!fps_skipped(x, y, z)) {
      const int pos = atomicAdd(&cell_cnt[morton_cell(x, y, z, box, box + 3)], 1);
      sp[pos] = make_float4(x, y, z, __builtin_bit_cast(float, k));
    }
  }
  for (int t = n_valid + tid; t < n_buckets * kWave; t += kThreads)
    sp[t] = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, -1));
  if (tid == 0) out[0] = 0;
  __syncthreads();  // scratch writes of this workgroup are visible to its own waves

  if (n_valid == 0) {  // every point skipped: the reference keeps returning index 0
    for (int j = 1 + tid; j < m; j += kThreads) out[j] = 0;
    return;
  }

  // ---- P4: per-bucket state ---------------------------------------------------------------
  // bucket jj of the wave <-> metadata set jj / 64 of lane jj % 64 (one set up to 4096 points
  // per wave, two sets beyond: 65 536 < N <= 81 920)
  constexpr int META = (NBW + kWave - 1) / kWave;
  const float4 *wave_base = sp + w * kWave;          // bucket (w + 16 jj) starts 16 KiB * jj later
  const unsigned lane_off = (unsigned)lane * (unsigned)sizeof(float4);
  float td[NBW];                       // running distance of MY point of bucket jj
  float blx[META], bly[META], blz[META], bhx[META], bhy[META], bhz[META];  // bucket `lane` of set s
  float bval[META], bx[META], by[META], bz[META];
  int bidx[META];
#pragma unroll
  for (int s = 0; s < META; ++s) {
    blx[s] = bly[s] = blz[s] = bhx[s] = bhy[s] = bhz[s] = 0.f;
    bval[s] = -2.0f; bx[s] = by[s] = bz[s] = 0.f; bidx[s] = 0;
  }
#pragma clang loop unroll(full)
  for (int jj = 0; jj < NBW; ++jj) {
    const int bid = w + kWaves * jj;
    td[jj] = -1.0f;
    if (bid < n_buckets) {  // wave-uniform
      const float4 q = load_bucket(wave_base, jj, lane_off);
      const bool real = __builtin_bit_cast(int, q.w) >= 0;
      td[jj] = real ? 1e10f : -1.0f;
      const float lx = wave_min_f32(real ? q.x : 3.0e38f), hx = wave_max_f32(real ? q.x : -3.0e38f);
      const float ly = wave_min_f32(real ? q.y : 3.0e38f), hy = wave_max_f32(real ? q.y : -3.0e38f);
      const float lz = wave_min_f32(real ? q.z : 3.0e38f), hz = wave_max_f32(real ? q.z : -3.0e38f);
      if (lane == jj % kWave) {
        blx[jj / kWave] = lx; bly[jj / kWave] = ly; blz[jj / kWave] = lz;
        bhx[jj / kWave] = hx; bhy[jj / kWave] = hy; bhz[jj / kWave] = hz;
        bval[jj / kWave] = 1e10f;  // forces the first round to visit the bucket
      }
    }
  }

  // ---- rounds -----------------------------------------------------------------------------
  float x1 = pts[0], y1 = pts[1], z1 = pts[2];
  for (int j = 1; j < m; ++j) {
    // (1) which of my wave's buckets can change?  lane jj % 64 answers for bucket jj
    unsigned long long visit[META];
#pragma unroll
    for (int s = 0; s < META; ++s) {
      const float lb = box_dist2(x1, y1, z1, blx[s], bly[s], blz[s], bhx[s], bhy[s], bhz[s]);
      visit[s] = __ballot(bval[s] > -2.0f && lb * 0.999999f < bval[s]);
    }
    // (2) update the selected buckets
#pragma clang loop unroll(full)
    for (int jj = 0; jj < NBW; ++jj) {
      if ((visit[jj / kWave] >> (jj % kWave)) & 1ull) {  // wave-uniform
        const float4 q = load_bucket(wave_base, jj, lane_off);
        const float d = sqdist3(q.x, q.y, q.z, x1, y1, z1);
        const float d2 = fminf(d, td[jj]);
        td[jj] = d2;
        const int qi = __builtin_bit_cast(int, q.w);
        const int win = wave_argmax_lane(d2, qi, log2bs);
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d2), win));
        const int vi = __builtin_amdgcn_readlane(qi, win);
        const float vx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.x), win));
        const float vy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.y), win));
        const float vz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.z), win));
        if (lane == jj % kWave) {
          bval[jj / kWave] = v; bidx[jj / kWave] = vi;
          bx[jj / kWave] = vx; by[jj / kWave] = vy; bz[jj / kWave] = vz;
        }
      }
    }
    // (3) best bucket of the lane's sets (value, then the reference's tie order), of the wave,
    //     then of the workgroup
    float cv = bval[0], cx = bx[0], cy = by[0], cz = bz[0];
    int ci = bidx[0];
#pragma unroll
    for (int s = 1; s < META; ++s) {
      const bool better = bval[s] > cv ||
                          (bval[s] == cv && fps_key(bidx[s], log2bs) < fps_key(ci, log2bs));
      if (better) { cv = bval[s]; ci = bidx[s]; cx = bx[s]; cy = by[s]; cz = bz[s]; }
    }
    const FpsPick p = fps_block_pick<kWaves>(cv, ci, cx, cy, cz, slots[j & 1], log2bs);
    if (p.idx == 0) { x1 = pts[0]; y1 = pts[1]; z1 = pts[2]; } else { x1 = p.x; y1 = p.y; z1 = p.z; }
    if (tid == 0) out[j] = p.idx;
  }
}

}  // namespace

// largest cloud the bucketed tier accepts: 16 waves x 80 buckets x 64 points (the running
// distances are one VGPR per owned point: 80 of the 128 a 1024-lane workgroup may use)
constexpr int kBucketMaxPoints = kThreads * 80;

size_t pn2_fps_bucket_scratch_bytes(int b, int n) {
  if (n > kBucketMaxPoints) return 0;
  const int nbw = (n + kThreads - 1) / kThreads;
  int tier = 8;
  while (tier < nbw) tier += 8;
  return sizeof(float4) * (size_t)b * kThreads * tier;
}

// returns 0 and sets *handled when the bucketed kernel was launched
// largest cloud whose cell lists the kernel can emit (16-bit scatter cursors)
int pn2_fps_bucket_grid_max_points() { return 65535; }

int pn2_fps_bucket_try(int b, int n, int m, int log2bs, const float *dataset, void *scratch,
                       size_t scratch_bytes, int *idxs, hipStream_t stream, int *handled,
                       float grid_radius, void *grid) {
  *handled = 0;
  if (n > kBucketMaxPoints || scratch == nullptr) return 0;
  if (scratch_bytes < pn2_fps_bucket_scratch_bytes(b, n)) return 0;
  const int nbw = (n + kThreads - 1) / kThreads;
  float4 *sc = reinterpret_cast<float4 *>(scratch);
  int *g_start = nullptr;
  float4 *g_rec = nullptr;
  float g_inv = 0.f;
  if (grid != nullptr) {  // also leave the cell lists for ball queries of grid_radius behind
    if (n > pn2_fps_bucket_grid_max_points() || n < 4096) return (int)hipErrorInvalidValue;
    const grid::GridWs ws = grid::grid_ws_layout(grid, b, n);
    g_start = ws.start;
    g_rec = ws.rec;
    g_inv = grid::grid_inv_side(grid_radius);
  }
#define FPS_BUCKET(T)                                                                         \
  hipLaunchKernelGGL((fps_bucket_kernel<T>), dim3(b), dim3(kThreads), 0, stream, n, m, log2bs, \
                     dataset, sc, idxs, g_inv, g_start, g_rec)
  if (nbw <= 8) FPS_BUCKET(8);
  else if (nbw <= 16) FPS_BUCKET(16);
  else if (nbw <= 24) FPS_BUCKET(24);
  else if (nbw <= 32) FPS_BUCKET(32);
  else if (nbw <= 40) FPS_BUCKET(40);
  else if (nbw <= 48) FPS_BUCKET(48);
  else if (nbw <= 56) FPS_BUCKET(56);
  else if (nbw <= 64) FPS_BUCKET(64);
  else if (nbw <= 72) FPS_BUCKET(72);
  else FPS_BUCKET(80);
#undef FPS_BUCKET
  *handled = 1;
  return pn2_launch_status();
}
