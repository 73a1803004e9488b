// 3dioumatch_amd/csrc/pn2_fps_bucket.hip -- exact furthest point sampling for LARGE clouds
// (SA1: 40 000 -> 2048), one workgroup per cloud, with spatial pruning.
//
// Why: FPS is m-1 strictly dependent rounds.  The reference (sampling_gpu.cu:75-178) and the
// streaming tier of pn2_sampling.hip re-read every point every round (N*16 bytes per round
// from L2: 8.9 us per round at N = 40 000, 18 ms per call -- 40 % of a train step).  But a new
// sample only lowers the running distance of points that are closer to it than their current
// distance, and those are spatially clustered.  So:
//
//   * SETUP kernel (1024 lanes per cloud): the cloud is counting-sorted by a coarse Morton cell
//     (16^3 cells, LDS histogram + LDS atomic cursors) into records (x, y, z, RUNNING DISTANCE)
//     + an array of original indices, cut into BUCKETS of 64 consecutive sorted points, each
//     with its tight bounding box.  A second workgroup per cloud builds, on request, the
//     ball-query cell lists of the same cloud (grid_common.h);
//   * ROUNDS kernel (8 waves per cloud -- two per SIMD).  Bucket b belongs to wave b % 8; lane j
//     of a wave holds box + current maximum of the wave's buckets j, 64 + j, 128 + j;
//   * a round = (1) every lane tests its buckets: if the squared distance from the new sample
//     to the bounding box (shrunk by 1e-6 relative, to stay conservative under fp32 rounding)
//     is not below the bucket's maximum, no point of the bucket can change -> skip;
//     (2) the wave walks the set bits of the ballots, two buckets at a time: one 1 KiB load per
//     bucket (up to three pairs in flight), the reference's exact fp32 update, a 4-byte store
//     of the new distances, both 64-lane maxima from one reduction tree; the lane that owns a
//     bucket's farthest point writes it to LDS;
//     (3) the wave's maximum, an 8-byte exchange across the waves through LDS with one
//     barrier, the winner's coordinates from LDS.
//   After a few dozen rounds ~15 of the 625 buckets of a 40 000-point cloud are touched per round.
//
// Why this shape: a round is a latency chain, not a throughput problem.  Measured with per-round
// clocks (tools/micro/fps_probe.py, profiles/r3_fps_round_clocks.json) the round-2 form -- one
// kernel, 16 waves, the wave's 40 buckets as 40 unrolled wave-uniform branches, the running
// distances in 40 registers per lane -- spent 1400 of its 5000 clocks per round walking the
// branch ladder (84 KB of code), 1900 in the pick (every one of the 16 waves repeats the final
// argmax; four waves share a SIMD's issue slot) and 750 per visited bucket with the loads
// serialised.  The rounds kernel below prices every construct on the chain
// (profiles/r3_instruction_costs.json): 1.27 us per round instead of 2.15.
//
// Exactness: skipping is a no-op by construction (min(d, t) == t for every point of a skipped
// bucket), the update arithmetic is the reference's, and ties are resolved by the reference's
// reduction-tree order through fps_key -- so the arrangement of points in buckets (which
// depends on atomic ordering) cannot influence the result.
#include "common.h"
#include "fps_common.h"
#include "grid_common.h"

#ifdef FPS_PROBE  // tools/micro/fps_probe.py: per-round clocks and visit counts of cloud 0
__device__ unsigned long long fps_probe_t[2048 * 16 * 16];
__device__ unsigned char fps_probe_v[2048 * 16];
#define FPS_STAMP(I)                                                              \
  if (blockIdx.x == 0 && j < 2048) {                                              \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                   \
    if (lane == 0) fps_probe_t[(j * 16 + w) * 16 + (I)] = t_;                     \
  }
#else
#define FPS_STAMP(I)
#endif

// Raw buffer access (descriptor + per-lane offset + SCALAR offset).  Declared against the LLVM
// intrinsics: __builtin_amdgcn_raw_buffer_load_b128 of clang 22 / ROCm 7.2 returns element 0 in
// all four lanes of its result.
typedef int fps_i32x4 __attribute__((ext_vector_type(4)));
typedef float fps_f32x4 __attribute__((ext_vector_type(4)));
__device__ fps_f32x4 fps_buffer_load_x4(fps_i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void fps_buffer_store_f32(float data, fps_i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.store.f32");

namespace {

using namespace fps;

__device__ __forceinline__ fps_i32x4 buffer_rsrc(const void *p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  fps_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));   // stride 0: raw buffer
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;                                                               // 32-bit data format
  return r;
}

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / kWave;  // 16
constexpr int kGrid = 16;                 // cells per axis
constexpr int kCells = kGrid * kGrid * kGrid;

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

// ---- setup: sort the cloud into buckets (blockIdx.y == 0) / build its cell lists (== 1) -----
__global__ void __launch_bounds__(kThreads)
fps_bucket_setup_kernel(int n, size_t cloud_stride, const float *__restrict__ dataset,
                        float4 *__restrict__ rec_all, int *__restrict__ pidx_all,
                        float *__restrict__ bbox_all, int *__restrict__ n_valid_out,
                        float grid_inv_side, int *__restrict__ grid_start,
                        float4 *__restrict__ grid_rec) {
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int w = tid / kWave;
  const float *pts = dataset + (size_t)blockIdx.x * n * 3;

  // ---- G: the ball-query cell lists of this cloud (grid_common.h) -------------------------
  // The set-abstraction layer that samples this cloud queries balls in it right afterwards:
  // 32768 16-bit counters packed in 64 KB of LDS (n < 65536 on this path), scan, scatter.  The
  // separate two-kernel build of the ball query disappears from the layer.
  if (blockIdx.y == 1) {
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
      if (tid == kThreads - 1) {
        grid_start[(size_t)blockIdx.x * grid::kStartStride + grid::kCells] = run;
        grid_start[(size_t)blockIdx.x * grid::kStartStride + grid::kOrderFor] = 0;  // no launch order yet
      }
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
    return;
  }

  __shared__ int cell_cnt[kCells];                 // histogram, then scatter cursors
  __shared__ float red[kWaves * 8];
  __shared__ float box[8];                         // lo[3], inv[3]
  __shared__ int s_valid;
  float4 *sp = rec_all + (size_t)blockIdx.x * cloud_stride;      // (x, y, z, running distance)
  int *spi = pidx_all + (size_t)blockIdx.x * cloud_stride;        // original index
  float *sbox = bbox_all + (size_t)blockIdx.x * (cloud_stride / kWave) * 8;

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
      sp[pos] = make_float4(x, y, z, 1e10f);
      spi[pos] = k;
    }
  }
  for (int t = n_valid + tid; t < n_buckets * kWave; t += kThreads) {
    sp[t] = make_float4(0.f, 0.f, 0.f, -1.0f);   // below every real distance: never picked
    spi[t] = -1;
  }
  if (tid == 0) n_valid_out[blockIdx.x] = n_valid;
  __syncthreads();  // scratch writes of this workgroup are visible to its own waves

  // ---- P4: tight bounding box of every bucket ---------------------------------------------
  for (int bkt = w; bkt < n_buckets; bkt += kWaves) {
    const float4 q = sp[bkt * kWave + lane];
    const bool real = q.w >= 0.f;
    const float lx = wave_min_f32(real ? q.x : 3.0e38f), hx = wave_max_f32(real ? q.x : -3.0e38f);
    const float ly = wave_min_f32(real ? q.y : 3.0e38f), hy = wave_max_f32(real ? q.y : -3.0e38f);
    const float lz = wave_min_f32(real ? q.z : 3.0e38f), hz = wave_max_f32(real ? q.z : -3.0e38f);
    if (lane == 0) {
      *reinterpret_cast<float4 *>(sbox + bkt * 8) = make_float4(lx, ly, lz, hx);
      *reinterpret_cast<float2 *>(sbox + bkt * 8 + 4) = make_float2(hy, hz);
    }
  }
}

// ---- rounds ---------------------------------------------------------------------------------
// W waves; bucket b belongs to wave b % W and is the wave's bucket jj = b / W; lane jj % 64 holds
// its bounding box and current maximum in metadata set jj / 64 (META sets); the point that
// attains the maximum (x, y, z, position in the sorted array) sits in LDS, written by the lane
// that owns that point.  Original indices are not touched during the rounds: the picks are
// recorded as positions and translated at the end (exact ties, which need the reference's
// index-based key, look the indices up on a slow path).
//
// A round is one dependent chain, and on this machine a dependent VALU instruction costs 8
// clocks, a DPP one 16, a VALU -> SGPR -> VALU round trip 32, any branch ~20, an LDS round trip
// ~110, an L2 hit ~220, and a vector-memory instruction occupies the CU's one address unit for
// 8..32 clocks (tools/micro/lat_probe.py, profiles/r3_instruction_costs.json).  So:
//   * up to G visited buckets are updated together: their 1 KiB loads are in flight at the same
//     time and their 64-lane maxima come out of ONE reduction tree (permlane swaps leave a part
//     of each bucket in each 16-lane row, then four DPP steps); when fewer than G remain the
//     last one is repeated -- the update is idempotent;
//   * addresses are buffer descriptor + per-lane offset + a SCALAR bucket offset: no VALU
//     address arithmetic;
//   * no value is moved between lanes: the lane that holds a bucket's farthest point writes it
//     to LDS itself, the bucket's maximum reaches its metadata lane as a scalar;
//   * the waves exchange (maximum, bucket id) only -- 8 bytes per wave -- and then read the
//     winner's coordinates from LDS;
//   * exact ties (between points of a bucket, buckets of a wave or waves) are detected with one
//     count each and resolved on a slow path by the reference's reduction-tree key.
template <int W, int META, int G>
__global__ void __launch_bounds__(W * kWave)
fps_bucket_rounds_kernel(int n, int m, int log2bs, size_t cloud_stride,
                         const float *__restrict__ dataset, float4 *__restrict__ rec_all,
                         const int *__restrict__ pidx_all, const float *__restrict__ bbox_all,
                         const int *__restrict__ n_valid_in, int *__restrict__ idxs,
                         int *__restrict__ first_tie_out, const int *__restrict__ prefix_first_tie,
                         int *__restrict__ grid_start, int *__restrict__ grid_order,
                         int *__restrict__ grid_order_key, unsigned *__restrict__ grid_plan,
                         float grid_inv_side) {
  static_assert(G == 2 || G == 4, "group of 2 or 4 buckets");
  constexpr int NG = 3;   // groups whose loads are issued together
  __shared__ __attribute__((aligned(16))) int2 slots[2][W];   // (bits of the wave's maximum, bucket id)
  // by bucket id: (x, y, z, position) of its farthest point; then one scratch entry per lane, the
  // target of the lanes that have nothing to write (a select on the address instead of a branch)
  __shared__ float4 far_pt[W * META * kWave + W * kWave];

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int far_nowhere = W * META * kWave + tid;
  int *out = idxs + (size_t)blockIdx.x * m;
  const int *pidx = pidx_all + (size_t)blockIdx.x * cloud_stride;
  const int n_valid = n_valid_in[blockIdx.x];
  const int n_buckets = (n_valid + kWave - 1) / kWave;
  if (prefix_first_tie != nullptr && prefix_first_tie[blockIdx.x] >= n) {
    // the cloud is the head of a sampling sequence without ties so far: see pn2_hip.h
    for (int j = tid; j < m; j += W * kWave) out[j] = j;
    if (tid == 0 && first_tie_out != nullptr) first_tie_out[blockIdx.x] = prefix_first_tie[blockIdx.x];
    return;
  }
  if (tid == 0) out[0] = 0;
  if (n_valid == 0) {  // every point skipped: the reference keeps returning index 0
    for (int j = 1 + tid; j < m; j += W * kWave) out[j] = 0;
    if (tid == 0 && first_tie_out != nullptr) first_tie_out[blockIdx.x] = 0;
    return;
  }
  // Exact ties of the GLOBAL maximum are tracked on the side (bit 30 of a bucket's far-point
  // position: its maximum is held by two lanes; bit 30 of a wave's bucket id: two of its buckets
  // hold its maximum; equal wave maxima): first_tie = the first round in which two points were
  // equally far, or nothing was left to take.  Up to that round the picks are strict maxima over
  // ALL points, and distinct.
  constexpr int kTieBit = 1 << 30, kNoTie = kTieBit - 1;
  int first_tie = m;
  // my point of the wave's jj-th bucket: descriptor + lane offset + jj * (W KiB)
  const fps_i32x4 rec_rs = buffer_rsrc(rec_all + (size_t)blockIdx.x * cloud_stride,
                                       (unsigned)(cloud_stride * sizeof(float4)));
  const int rec_off = (w * kWave + lane) * (int)sizeof(float4);
  constexpr int kBucketStep = W * kWave * (int)sizeof(float4);

  // ---- per-bucket state: bucket jj of the wave <-> metadata set jj / 64 of lane jj % 64 ------
  float blx[META], bly[META], blz[META], bhx[META], bhy[META], bhz[META], bval[META];
  {
    const float *sbox = bbox_all + (size_t)blockIdx.x * (cloud_stride / kWave) * 8;
#pragma unroll
    for (int s = 0; s < META; ++s) {
      blx[s] = bly[s] = blz[s] = bhx[s] = bhy[s] = bhz[s] = 0.f;
      bval[s] = -2.0f;
      const int bid = w + W * (s * kWave + lane);
      if (bid < n_buckets) {
        const float4 a = *reinterpret_cast<const float4 *>(sbox + bid * 8);
        const float2 c = *reinterpret_cast<const float2 *>(sbox + bid * 8 + 4);
        blx[s] = a.x; bly[s] = a.y; blz[s] = a.z; bhx[s] = a.w; bhy[s] = c.x; bhz[s] = c.y;
        bval[s] = 1e10f;  // forces the first round to visit the bucket
      }
    }
  }

  // ---- rounds -----------------------------------------------------------------------------
  const float *pts = dataset + (size_t)blockIdx.x * n * 3;
  float x1 = pts[0], y1 = pts[1], z1 = pts[2];
  for (int j = 1; j < m; ++j) {
    FPS_STAMP(0)
    FPS_STAMP(1)
#ifdef FPS_PROBE
    int probe_visits = 0;
    bool probe_first = true;
#endif
    // (1) which of my wave's buckets can change?  lane jj % 64 answers for bucket jj
    unsigned long long visit[META];
#pragma unroll
    for (int s = 0; s < META; ++s) {
      const float lb = box_dist2(x1, y1, z1, blx[s], bly[s], blz[s], bhx[s], bhy[s], bhz[s]);
      visit[s] = __ballot(lb * 0.999999f < bval[s]);   // (absent buckets: bval = -2 < 0 <= lb)
#ifdef FPS_PROBE
      probe_visits += __popcll(visit[s]);
#endif
    }
    FPS_STAMP(2)
    // (2) update the selected buckets: the loads of up to NG groups of G are issued together
    //     (one L2 latency per round, not per group), then group after group is reduced
#pragma unroll
    for (int s = 0; s < META; ++s) {
      unsigned long long vm = visit[s];
      while (vm) {
        const int nv = __popcll(vm);
        int bk[NG * G];
        fps_f32x4 q[NG * G];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g == 0 || nv > g * G) {   // wave-uniform
#pragma unroll
            for (int k = g * G; k < (g + 1) * G; ++k) {
              bk[k] = (k == 0 || vm) ? (int)__builtin_ctzll(vm) : bk[k - 1];
              vm = vm ? (vm & (vm - 1ull)) : 0ull;
              q[k] = fps_buffer_load_x4(rec_rs, rec_off, (s * kWave + bk[k]) * kBucketStep, 0);
            }
          }
        }
#ifdef FPS_PROBE
        if (probe_first) { FPS_STAMP(3) }
#endif
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g == 0 || nv > g * G) {   // wave-uniform
            const int k0 = g * G;
            float d2[G];
#pragma unroll
            for (int k = 0; k < G; ++k) {
              const float d = sqdist3(q[k0 + k].x, q[k0 + k].y, q[k0 + k].z, x1, y1, z1);
              const float told = q[k0 + k].w;
              asm("v_min_f32 %0, %1, %2" : "=v"(d2[k]) : "v"(d), "v"(told));
              fps_buffer_store_f32(d2[k], rec_rs, rec_off + 12, (s * kWave + bk[k0 + k]) * kBucketStep, 0);
            }
#ifdef FPS_PROBE
            if (probe_first) { FPS_STAMP(4) }
#endif
            // G maxima from one tree
            float r;
            float mx[G];
            {
              unsigned a0, a1;
              swap_rows<true>(__builtin_bit_cast(unsigned, d2[0]), __builtin_bit_cast(unsigned, d2[1]), a0, a1);
              float m01;
              {
                const float f0 = __builtin_bit_cast(float, a0), f1 = __builtin_bit_cast(float, a1);
                asm("v_max_f32 %0, %1, %2" : "=v"(m01) : "v"(f0), "v"(f1));
              }
              float m23 = m01;
              if (G == 4) {
                unsigned c0, c1;
                swap_rows<true>(__builtin_bit_cast(unsigned, d2[G - 2]), __builtin_bit_cast(unsigned, d2[G - 1]), c0, c1);
                const float g0 = __builtin_bit_cast(float, c0), g1 = __builtin_bit_cast(float, c1);
                asm("v_max_f32 %0, %1, %2" : "=v"(m23) : "v"(g0), "v"(g1));
              }
              swap_rows<false>(__builtin_bit_cast(unsigned, m01), __builtin_bit_cast(unsigned, m23), a0, a1);
              const float f0 = __builtin_bit_cast(float, a0), f1 = __builtin_bit_cast(float, a1);
              asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(f0), "v"(f1));
              FPS_DPP_OP("v_max_f32_dpp", r, "quad_perm:[1,0,3,2]");
              FPS_DPP_OP("v_max_f32_dpp", r, "quad_perm:[2,3,0,1]");
              FPS_DPP_OP("v_max_f32_dpp", r, "row_half_mirror");
              FPS_DPP_OP("v_max_f32_dpp", r, "row_mirror");
            }
            // rows 0..3 of r: G == 4: buckets 0, 2, 1, 3;  G == 2: buckets 0, 0, 1, 1
            mx[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 0));
            mx[1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 32));
            if (G == 4) {
              mx[G - 2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 16));
              mx[G - 1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 48));
            }
            unsigned long long tie[G];
            int holders = 0;   // lanes holding a maximum: G unless some bucket has an exact tie
#pragma unroll
            for (int k = 0; k < G; ++k) {
              tie[k] = __ballot(d2[k] == mx[k]);
              holders += __popcll(tie[k]);
            }
            int win[G];
#pragma unroll
            for (int k = 0; k < G; ++k) win[k] = __builtin_ctzll(tie[k]);
            if (__builtin_expect(holders > G, 0)) {  // exact ties inside a bucket: the reference's order decides
#pragma unroll
              for (int k = 0; k < G; ++k)
                if (tie[k] & (tie[k] - 1ull))
                  win[k] = wave_tie_break(tie[k], pidx[(w + W * (s * kWave + bk[k0 + k])) * kWave + lane], log2bs);
            }
#ifdef FPS_PROBE
            if (probe_first) { FPS_STAMP(5) }
#endif
#pragma unroll
            for (int k = 0; k < G; ++k) {
              const int gb = w + W * (s * kWave + bk[k0 + k]);
              const int pos0 = gb * kWave + ((tie[k] & (tie[k] - 1ull)) ? kTieBit : 0);
              far_pt[lane == win[k] ? gb : far_nowhere] =
                  make_float4(q[k0 + k].x, q[k0 + k].y, q[k0 + k].z, __builtin_bit_cast(float, pos0 + lane));
              bval[s] = lane == bk[k0 + k] ? mx[k] : bval[s];
            }
#ifdef FPS_PROBE
            if (probe_first) { FPS_STAMP(6) }
            probe_first = false;
#endif
          }
        }
      }
    }
    FPS_STAMP(7)
#ifdef FPS_PROBE
    if (blockIdx.x == 0 && lane == 0 && j < 2048)
      fps_probe_v[j * 16 + w] = (unsigned char)(probe_visits > 255 ? 255 : probe_visits);
#endif
    // (3) the wave's farthest bucket ...
    float cv = bval[0];
#pragma unroll
    for (int s = 1; s < META; ++s) cv = fmaxf(cv, bval[s]);
    const float wm = wave_max_f32(cv);
    FPS_STAMP(8)
    unsigned long long eq[META];
    int total = 0;
#pragma unroll
    for (int s = 0; s < META; ++s) {
      eq[s] = __ballot(bval[s] == wm);
      total += __popcll(eq[s]);
    }
    int best_jj = 0;  // wave-uniform: the wave's bucket holding its farthest point
#pragma unroll
    for (int s = META - 1; s >= 0; --s)
      if (eq[s]) best_jj = s * kWave + (int)__builtin_ctzll(eq[s]);
    if (__builtin_expect(total > 1 && wm > -2.0f, 0)) {  // equal maxima in several buckets: smallest key wins
      unsigned key = 0xFFFFFFFFu;
      int kjj = 0;
#pragma unroll
      for (int s = 0; s < META; ++s) {
        if (bval[s] == wm) {
          const int jj = s * kWave + lane;
          const unsigned kk = fps_key(pidx[__builtin_bit_cast(int, far_pt[w + W * jj].w) & kNoTie], log2bs);
          if (kk < key) { key = kk; kjj = jj; }
        }
      }
      const unsigned mk = wave_min_u32(key);
      best_jj = __builtin_amdgcn_readlane(kjj, __builtin_ctzll(__ballot(key == mk)));
    }
    FPS_STAMP(9)
    if (lane == 0)
      slots[j & 1][w] = make_int2(__builtin_bit_cast(int, wm),
                                  (w + W * best_jj) | ((total > 1 && wm > -2.0f) ? kTieBit : 0));
    __syncthreads();
    FPS_STAMP(10)
    // ... and the workgroup's: lane i < W takes wave i's candidate, log2(W) DPP steps, one ballot
    constexpr int CL = W <= 8 ? 8 : 16;
    int2 cand = make_int2(__builtin_bit_cast(int, -2.0f), 0);
    if (lane < W) cand = slots[j & 1][lane];
    const float cvl = __builtin_bit_cast(float, cand.x);
    const float best = wave_max_f32<CL>(cvl);
    const unsigned long long ceq = __ballot(cvl == best) & ((1ull << W) - 1ull);
    int pick_gb = __builtin_amdgcn_readlane(cand.y, (int)__builtin_ctzll(ceq));
    bool tied = (ceq & (ceq - 1ull)) != 0ull;
    if (__builtin_expect(tied, 0)) {  // equal maxima in several waves: smallest key wins
      unsigned key = 0xFFFFFFFFu;
      if ((ceq >> lane) & 1ull)
        key = fps_key(pidx[__builtin_bit_cast(int, far_pt[cand.y & kNoTie].w) & kNoTie], log2bs);
      const unsigned mk = wave_min_u32(key);
      pick_gb = __builtin_amdgcn_readlane(cand.y, (int)__builtin_ctzll(__ballot(key == mk)));
    }
    // (a maximum of zero: every participating point has been taken -- from here on the picks
    //  repeat earlier ones, which is a tie for whoever samples them as a cloud)
    tied = tied || (pick_gb & kTieBit) != 0 || !(best > 0.f);
    pick_gb &= kNoTie;
    FPS_STAMP(11)
    const float4 c = far_pt[pick_gb];   // one address: broadcast read
    x1 = c.x; y1 = c.y; z1 = c.z;
    const int cpos = __builtin_bit_cast(int, c.w);
    if (tid == 0) out[j] = cpos & kNoTie;   // position; translated below
    if (tied || (cpos & kTieBit) != 0) first_tie = first_tie < j ? first_tie : j;
    __syncthreads();   // the owner of pick_gb rewrites far_pt[pick_gb] in the next round
    FPS_STAMP(12)
  }
  // positions -> original indices (wave 0 wrote them: same-wave program order)
  if (w == 0)
    for (int j = 1 + lane; j < m; j += kWave) out[j] = pidx[out[j]];
  if (tid == 0 && first_tie_out != nullptr) first_tie_out[blockIdx.x] = first_tie;

  // ---- launch order for the ball queries around these picks (grid_common.h) ------------------
  // The set-abstraction layer queries a ball around every pick right afterwards, one wave per
  // centroid.  A centroid inside a dense cluster sweeps up to ~30 chunks of candidates (10 us
  // against 3 for the usual one); dispatched last it is the tail of that launch.  The cell lists
  // of the cloud are in place (the setup kernel's second workgroup), so the cost class of every
  // query is 18 cached loads away: a counting sort by class, longest first -- the query kernel's
  // workgroup jj answers centroid order[jj], and the long ones run under everyone else.
  if (grid_order != nullptr && m <= n) {   // kernel arguments
    __shared__ int o_cnt[kWave];
    const int *st = grid_start + (size_t)blockIdx.x * grid::kStartStride;
    int *ord = grid_order + (size_t)blockIdx.x * n;
    int *key = grid_order_key + (size_t)blockIdx.x * n;
    if (tid < kWave) o_cnt[tid] = 0;
    __syncthreads();   // (also: wave 0's translated picks)
    for (int jj = tid; jj < m; jj += W * kWave) {
      const int p = out[jj];
      const int k = grid::query_cost_class(st, pts[p * 3 + 0], pts[p * 3 + 1], pts[p * 3 + 2], grid_inv_side);
      key[jj] = k;
      atomicAdd(&o_cnt[k], 1);
    }
    __syncthreads();
    if (w == 0) {   // exclusive scan from the longest class down: lane l <-> class 63 - l
      const int c = o_cnt[kWave - 1 - lane];
      int incl = c;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
      }
      o_cnt[kWave - 1 - lane] = incl - c;
    }
    __syncthreads();
    // ... and, at the same position, the query's PLAN (grid_common.h): row offsets, row lengths,
    // the centroid itself -- the query wave's whole head as one 64-byte load
    const int plan_cap = grid::grid_plan_capacity(n);
    unsigned *plan = grid_plan != nullptr && m <= plan_cap && m < 65536
                         ? grid_plan + (size_t)blockIdx.x * plan_cap * grid::kPlanWords : nullptr;
    for (int jj = tid; jj < m; jj += W * kWave) {
      const int pos = atomicAdd(&o_cnt[key[jj]], 1);
      ord[pos] = jj;
      if (plan != nullptr) {
        const int p = out[jj];
        grid::write_query_plan(plan + (size_t)pos * grid::kPlanWords, st, pts[p * 3 + 0], pts[p * 3 + 1],
                               pts[p * 3 + 2], grid_inv_side, jj, m, key[jj]);
      }
    }
    if (tid == 0) grid_start[(size_t)blockIdx.x * grid::kStartStride + grid::kOrderFor] = m;
  }
}

}  // namespace

#ifndef FPS_BUCKET_WAVES
#define FPS_BUCKET_WAVES 8
#endif
#ifndef FPS_BUCKET_GROUP
#define FPS_BUCKET_GROUP 2
#endif

// largest cloud the bucketed tier accepts: W waves x 3 metadata sets x 64 buckets x 64 points
constexpr int kBucketMaxPoints = FPS_BUCKET_WAVES * 3 * kWave * kWave;

static size_t fps_bucket_cloud_stride(int n) {  // point slots per cloud: whole buckets
  return (size_t)((n + kWave - 1) / kWave) * kWave;
}

// per cloud: (x, y, z, distance) + index per slot, 8 floats of bounding box per bucket; then
// one int per cloud
size_t pn2_fps_bucket_scratch_bytes(int b, int n) {
  if (n > kBucketMaxPoints) return 0;
  const size_t stride = fps_bucket_cloud_stride(n);
  return (size_t)b * stride * (sizeof(float4) + sizeof(int)) + (size_t)b * (stride / kWave) * 8 * sizeof(float) +
         sizeof(int) * (((size_t)b + 63) & ~(size_t)63);
}

// largest cloud whose cell lists the setup kernel can emit (16-bit scatter cursors)
int pn2_fps_bucket_grid_max_points() { return 65535; }

// returns 0 and sets *handled when the bucketed kernels were launched
int pn2_fps_bucket_try(int b, int n, int m, int log2bs, const float *dataset, void *scratch,
                       size_t scratch_bytes, int *idxs, hipStream_t stream, int *handled,
                       float grid_radius, void *grid, int *first_tie_out, const int *prefix_first_tie) {
  *handled = 0;
  if (n > kBucketMaxPoints || scratch == nullptr) return 0;
  if (scratch_bytes < pn2_fps_bucket_scratch_bytes(b, n)) return 0;
  const size_t stride = fps_bucket_cloud_stride(n);
  float4 *rec = reinterpret_cast<float4 *>(scratch);
  int *pidx = reinterpret_cast<int *>(rec + (size_t)b * stride);
  float *bbox = reinterpret_cast<float *>(pidx + (size_t)b * stride);
  int *n_valid = reinterpret_cast<int *>(bbox + (size_t)b * (stride / kWave) * 8);
  int *g_start = nullptr, *g_order = nullptr, *g_order_key = nullptr;
  unsigned *g_plan = nullptr;
  float4 *g_rec = nullptr;
  float g_inv = 0.f;
  if (grid != nullptr) {  // also leave the cell lists for ball queries of grid_radius behind
    if (n > pn2_fps_bucket_grid_max_points() || n < 4096) return (int)hipErrorInvalidValue;
    const grid::GridWs ws = grid::grid_ws_layout(grid, b, n);
    g_start = ws.start;
    g_rec = ws.rec;
    g_order = ws.order;
    g_order_key = ws.order_key;
    g_plan = ws.plan;
    g_inv = grid::grid_inv_side(grid_radius);
  }
  hipLaunchKernelGGL(fps_bucket_setup_kernel, dim3(b, grid != nullptr ? 2 : 1), dim3(kThreads), 0, stream,
                     n, stride, dataset, rec, pidx, bbox, n_valid, g_inv, g_start, g_rec);
  constexpr int WV = FPS_BUCKET_WAVES;
  const int per_wave = ((int)(stride / kWave) + WV - 1) / WV;   // buckets per wave
#define FPS_ROUNDS(META)                                                                       \
  hipLaunchKernelGGL((fps_bucket_rounds_kernel<WV, META, FPS_BUCKET_GROUP>), dim3(b), dim3(WV * kWave), 0, \
                     stream, n, m, log2bs, stride, dataset, rec, pidx, bbox, n_valid, idxs, first_tie_out, prefix_first_tie,  \
                     g_start, g_order, g_order_key, g_plan, g_inv)
  if (per_wave <= kWave) FPS_ROUNDS(1);
  else if (per_wave <= 2 * kWave) FPS_ROUNDS(2);
  else FPS_ROUNDS(3);
#undef FPS_ROUNDS
  *handled = 1;
  return pn2_launch_status();
}
