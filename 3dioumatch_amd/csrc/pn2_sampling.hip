// 3dioumatch_amd/csrc/pn2_sampling.hip -- furthest point sampling + index gather for gfx950.
//
// Semantics: reference pointnet2/_ext_src/src/sampling_gpu.cu (K1 :75-234, K2 :13-35,
// K3 :39-62), restated in SURVEY App. A.1/A.2.  Design is new:
//
//  * FPS.  The reference runs ONE 512-thread block per cloud and re-streams xyz+temp from
//    global memory every round.  Here a cloud's points live in VGPRs for the whole call
//    (x, y, z, running min distance: 16 B/point, up to 16 points per lane at 1024 lanes),
//    a round is: per-lane update -> 64-lane wave argmax (cross-lane shuffles) -> one LDS
//    slot per wave -> ONE workgroup barrier (double-buffered slots) -> every wave redundantly
//    reduces the <=16 slots.  Clouds larger than the register budget fall back to a
//    streaming variant (temp in the caller's scratch buffer).
//  * Exact index parity.  The reference's result depends on its reduction tree: among equal
//    maxima the winner minimises bitreverse(k mod bs) and then k, where bs = 2^floor(log2 n)
//    capped at 512 (cuda_utils.h:20-24).  That order is reproduced by comparing
//    (value, key(k)) with key = bitrev(k mod bs) << 22 | k, independent of how many lanes
//    or waves this kernel uses.
//  * Points with |p|^2 <= 1e-3 never take part (sampling_gpu.cu:105-106); they are given a
//    running distance of -1, which can neither be selected (best starts at -1, strict >)
//    nor change (min(d, -1) = -1).
#include <stdlib.h>

#include "common.h"
#include "fps_common.h"

namespace {

using namespace fps;

// Register-resident FPS: THREADS lanes, each owning points tid + i*THREADS, i < PPT.
// THREADS is a multiple of the reference block size bs, so k mod bs is constant per lane
// and the lane-local scan in ascending i already honours the tie order.
template <int THREADS, int PPT>
__global__ void __launch_bounds__(THREADS)
fps_reg_kernel(int n, int m, int log2bs, const float *__restrict__ dataset,
               int *__restrict__ idxs, const int *__restrict__ prefix_first_tie) {
  constexpr int NW = THREADS / kWave;
  __shared__ __attribute__((aligned(16))) float slots[2][NW * 8];
  const int tid = threadIdx.x;
  const float *pts = dataset + (size_t)blockIdx.x * n * 3;
  int *out = idxs + (size_t)blockIdx.x * m;
  if (prefix_first_tie != nullptr && prefix_first_tie[blockIdx.x] >= n) {
    // the cloud is the head of a sampling sequence without ties so far (pn2_hip.h)
    for (int j = tid; j < m; j += THREADS) out[j] = j;
    return;
  }

  float px[PPT], py[PPT], pz[PPT], td[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * THREADS;
    if (k < n) {
      px[i] = pts[k * 3 + 0];
      py[i] = pts[k * 3 + 1];
      pz[i] = pts[k * 3 + 2];
      td[i] = fps_skipped(px[i], py[i], pz[i]) ? -1.0f : 1e10f;
    } else {
      px[i] = py[i] = pz[i] = 0.f;
      td[i] = -1.0f;
    }
  }
  if (tid == 0) out[0] = 0;
  float x1 = pts[0], y1 = pts[1], z1 = pts[2];
  for (int j = 1; j < m; ++j) {
    float best = -1.0f, bx = x1, by = y1, bz = z1;
    int besti = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = sqdist3(px[i], py[i], pz[i], x1, y1, z1);
      const float d2 = fminf(d, td[i]);
      td[i] = d2;
      if (d2 > best) { best = d2; besti = tid + i * THREADS; bx = px[i]; by = py[i]; bz = pz[i]; }
    }
    const FpsPick p = fps_block_pick<NW>(best, besti, bx, by, bz, slots[j & 1], log2bs);
    // no candidate anywhere (every point skipped): the reference re-reads point 0
    if (p.idx == 0) { x1 = pts[0]; y1 = pts[1]; z1 = pts[2]; } else { x1 = p.x; y1 = p.y; z1 = p.z; }
    if (tid == 0) out[j] = p.idx;
  }
}

// Streaming FPS for clouds beyond the register budget: xyz and the running distances are
// re-read from memory (L2-resident) every round, exactly the reference's data flow but with
// 1024 lanes.  THREADS (1024) is a multiple of bs (<= 512), see above.
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
fps_stream_kernel(int n, int m, int log2bs, const float *__restrict__ dataset,
                  float *__restrict__ temp, int *__restrict__ idxs,
                  const int *__restrict__ prefix_first_tie) {
  constexpr int NW = THREADS / kWave;
  __shared__ __attribute__((aligned(16))) float slots[2][NW * 8];
  const int tid = threadIdx.x;
  const float *pts = dataset + (size_t)blockIdx.x * n * 3;
  float *tmp = temp + (size_t)blockIdx.x * n;
  int *out = idxs + (size_t)blockIdx.x * m;
  if (prefix_first_tie != nullptr && prefix_first_tie[blockIdx.x] >= n) {
    for (int j = tid; j < m; j += THREADS) out[j] = j;
    return;
  }

  for (int k = tid; k < n; k += THREADS)
    tmp[k] = fps_skipped(pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2]) ? -1.0f : 1e10f;
  if (tid == 0) out[0] = 0;
  float x1 = pts[0], y1 = pts[1], z1 = pts[2];
  for (int j = 1; j < m; ++j) {
    float best = -1.0f, bx = x1, by = y1, bz = z1;
    int besti = 0;
    for (int k = tid; k < n; k += THREADS) {
      const float x = pts[k * 3 + 0], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
      const float d = sqdist3(x, y, z, x1, y1, z1);
      const float d2 = fminf(d, tmp[k]);
      tmp[k] = d2;
      if (d2 > best) { best = d2; besti = k; bx = x; by = y; bz = z; }
    }
    const FpsPick p = fps_block_pick<NW>(best, besti, bx, by, bz, slots[j & 1], log2bs);
    if (p.idx == 0) { x1 = pts[0]; y1 = pts[1]; z1 = pts[2]; } else { x1 = p.x; y1 = p.y; z1 = p.z; }
    if (tid == 0) out[j] = p.idx;
  }
}

// out[b,c,j] = points[b,c,idx[b,j]]   (sampling_gpu.cu:13-25)
__global__ void __launch_bounds__(256)
gather_points_kernel(int c, int n, int m, const float *__restrict__ points,
                     const int *__restrict__ idx, float *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const int b = blockIdx.z;
  const int a = idx[(size_t)b * m + j];
  for (int l = blockIdx.y; l < c; l += gridDim.y)
    out[((size_t)b * c + l) * m + j] = points[((size_t)b * c + l) * n + a];
}

// grad_points[b,c,idx[b,j]] += grad_out[b,c,j]   (sampling_gpu.cu:39-52)
__global__ void __launch_bounds__(256)
gather_points_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                          const int *__restrict__ idx, float *__restrict__ grad_points) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const int b = blockIdx.z;
  const int a = idx[(size_t)b * m + j];
  for (int l = blockIdx.y; l < c; l += gridDim.y)
    atomicAdd(grad_points + ((size_t)b * c + l) * n + a, grad_out[((size_t)b * c + l) * m + j]);
}

int ref_log2_block(int n) {
  // log2 of the reference's block size: opt_n_threads(n), cuda_utils.h:20-24
  int l = 0;
  while ((2 << l) <= n && l < 9) ++l;
  return l;
}

}  // namespace

size_t pn2_fps_bucket_scratch_bytes(int b, int n);
int pn2_fps_bucket_grid_max_points();
size_t pn2_grid_layout_bytes(int b, int n);
int pn2_fps_bucket_try(int b, int n, int m, int log2bs, const float *dataset, void *scratch,
                       size_t scratch_bytes, int *idxs, hipStream_t stream, int *handled,
                       float grid_radius, void *grid, int *first_tie_out, const int *prefix_first_tie);

// clouds with at least this many points use the bucketed (spatially pruned) tier when a
// workspace is supplied; overridable for experiments with PN2_FPS_BUCKET_MIN_N
static int fps_bucket_min_n() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("PN2_FPS_BUCKET_MIN_N");
    v = e ? atoi(e) : 8192;
    if (v < 1) v = 1;
  }
  return v;
}

PN2_API size_t pn2_fps_workspace_bytes(int b, int n, int m) {
  (void)m;
  if (n >= fps_bucket_min_n()) {
    const size_t s = pn2_fps_bucket_scratch_bytes(b, n);
    if (s) return s;
  }
  return n > 16384 ? sizeof(float) * (size_t)b * n : 0;  // streaming tier: (b,n) distances
}

// every tier; prefix_first_tie (device, per cloud, or null): see pn2_furthest_point_sampling_prefix
static int fps_dispatch(int b, int n, int m, const float *dataset, int *idxs, void *workspace,
                        size_t workspace_bytes, const int *prefix_first_tie, hipStream_t stream) {
  if (b <= 0 || m <= 0) return 0;
  if (n <= 0 || n >= (1 << 22)) return (int)hipErrorInvalidValue;
  const int log2bs = ref_log2_block(n);
  if (n >= fps_bucket_min_n()) {
    int handled = 0;
    const int rc = pn2_fps_bucket_try(b, n, m, log2bs, dataset, workspace, workspace_bytes, idxs,
                                      stream, &handled, 0.f, nullptr, nullptr, prefix_first_tie);
    if (rc != 0 || handled) return rc;
  }
#define FPS_REG(T, P)                                                                     \
  hipLaunchKernelGGL((fps_reg_kernel<T, P>), dim3(b), dim3(T), 0, stream, n, m, log2bs,   \
                     dataset, idxs, prefix_first_tie)
  if (n < 512) {            // bs <= 256 divides 256
    FPS_REG(256, 2);
  } else if (n <= 512) {
    FPS_REG(512, 1);
  } else if (n <= 1024) {
    FPS_REG(512, 2);
  } else if (n <= 2048) {
    FPS_REG(512, 4);
  } else if (n <= 4096) {
    FPS_REG(1024, 4);
  } else if (n <= 8192) {
    FPS_REG(1024, 8);
  } else if (n <= 16384) {
    FPS_REG(1024, 16);
  } else {
    if (!workspace || workspace_bytes < sizeof(float) * (size_t)b * n)
      return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(b), dim3(1024), 0, stream, n, m, log2bs,
                       dataset, (float *)workspace, idxs, prefix_first_tie);
  }
#undef FPS_REG
  return pn2_launch_status();
}

PN2_API int pn2_furthest_point_sampling_ws(int b, int n, int m, const float *dataset,
                                           int *idxs, void *workspace, size_t workspace_bytes,
                                           void *stream_) {
  return fps_dispatch(b, n, m, dataset, idxs, workspace, workspace_bytes, nullptr,
                      (hipStream_t)stream_);
}

PN2_API int pn2_furthest_point_sampling_prefix(int b, int n, int m, const float *dataset, int *idxs,
                                               void *workspace, size_t workspace_bytes,
                                               const int *first_tie, void *stream_) {
  // more samples than points: the reference's sampling repeats picks (sampling_gpu.cu:94-177) and
  // the prefix shortcut does not describe that -- the plain sampling answers
  return fps_dispatch(b, n, m, dataset, idxs, workspace, workspace_bytes, m > n ? nullptr : first_tie,
                      (hipStream_t)stream_);
}

PN2_API int pn2_fps_grid_supported(int n);

// Does pn2_furthest_point_sampling_ties accept clouds of n points?  (bucketed tier)
PN2_API int pn2_fps_ties_supported(int n) {
  return n >= fps_bucket_min_n() && pn2_fps_bucket_scratch_bytes(1, n) != 0;
}

PN2_API int pn2_furthest_point_sampling_ties(int b, int n, int m, const float *dataset, int *idxs,
                                             void *workspace, size_t workspace_bytes,
                                             float grid_radius, void *grid, size_t grid_bytes,
                                             int *first_tie, void *stream_) {
  if (b <= 0 || m <= 0) return 0;
  if (!pn2_fps_ties_supported(n) || !first_tie) return (int)hipErrorInvalidValue;
  if (grid != nullptr &&
      (!pn2_fps_grid_supported(n) || grid_bytes < pn2_grid_layout_bytes(b, n) ||
       !(grid_radius > 1e-6f) || !(grid_radius < 1e6f)))
    return (int)hipErrorInvalidValue;
  int handled = 0;
  const int rc = pn2_fps_bucket_try(b, n, m, ref_log2_block(n), dataset, workspace,
                                    workspace_bytes, idxs, (hipStream_t)stream_, &handled,
                                    grid_radius, grid, first_tie, nullptr);
  if (rc != 0) return rc;
  return handled ? 0 : (int)hipErrorInvalidValue;  // (workspace too small for the bucketed tier)
}

// Can pn2_furthest_point_sampling_grid leave cell lists behind for a cloud of n points?
PN2_API int pn2_fps_grid_supported(int n) {
  return n >= fps_bucket_min_n() && n >= 4096 && n <= pn2_fps_bucket_grid_max_points() &&
         pn2_grid_layout_bytes(1, n) != 0;
}

PN2_API int pn2_furthest_point_sampling_grid(int b, int n, int m, const float *dataset, int *idxs,
                                             void *workspace, size_t workspace_bytes,
                                             float grid_radius, void *grid, size_t grid_bytes,
                                             void *stream_) {
  if (b <= 0 || m <= 0) return 0;
  if (!pn2_fps_grid_supported(n) || !grid || grid_bytes < pn2_grid_layout_bytes(b, n) ||
      !(grid_radius > 1e-6f) || !(grid_radius < 1e6f))
    return (int)hipErrorInvalidValue;
  int handled = 0;
  const int rc = pn2_fps_bucket_try(b, n, m, ref_log2_block(n), dataset, workspace,
                                    workspace_bytes, idxs, (hipStream_t)stream_, &handled,
                                    grid_radius, grid, nullptr, nullptr);
  if (rc != 0) return rc;
  return handled ? 0 : (int)hipErrorInvalidValue;  // (workspace too small for the bucketed tier)
}

// Reference-shaped entry point: `temp` is the (b,n) float scratch of the reference ABI, which
// is enough for the register and streaming tiers (the bucketed tier needs the larger
// workspace of pn2_furthest_point_sampling_ws).
PN2_API int pn2_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                        int *idxs, void *stream_) {
  if (n > 16384 && !temp) return (int)hipErrorInvalidValue;
  // a (b,n) float buffer is too small for the bucketed tier, so that tier declines by itself
  return pn2_furthest_point_sampling_ws(b, n, m, dataset, idxs, temp,
                                        temp ? sizeof(float) * (size_t)b * n : 0, stream_);
}

PN2_API int pn2_gather_points(int b, int c, int n, int npoints, const float *points,
                              const int *idx, float *out, void *stream_) {
  if (b <= 0 || c <= 0 || npoints <= 0) return 0;
  (void)n;
  dim3 grid(pn2_ceil_div(npoints, 256), c < 65535 ? c : 65535, b);
  hipLaunchKernelGGL(gather_points_kernel, grid, dim3(256), 0, (hipStream_t)stream_, c, n,
                     npoints, points, idx, out);
  return pn2_launch_status();
}

PN2_API int pn2_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                   const int *idx, float *grad_points, void *stream_) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  const int e = pn2_zero_async(grad_points, sizeof(float) * (size_t)b * c * n, stream);
  if (e != 0) return e;
  if (npoints <= 0) return 0;
  dim3 grid(pn2_ceil_div(npoints, 256), c < 65535 ? c : 65535, b);
  hipLaunchKernelGGL(gather_points_grad_kernel, grid, dim3(256), 0, stream, c, n, npoints,
                     grad_out, idx, grad_points);
  return pn2_launch_status();
}

// ---- many small device-to-device copies in ONE launch -----------------------------------------
// The train step stages a batch (inputs + its prefetched index chain: ~46 tensors, 11 MB) from a
// prefetch slot into the buffers its captured graphs read; as 46 copy launches that is ~100 us of
// launch boundaries per step.  table: n rows of (src, dst, bytes) as 64-bit values on the device.
namespace {
__global__ void __launch_bounds__(256)
multi_copy_kernel(const unsigned long long *__restrict__ table) {
  const unsigned long long *row = table + (size_t)blockIdx.y * 3;
  const char *src = reinterpret_cast<const char *>(row[0]);
  char *dst = reinterpret_cast<char *>(row[1]);
  const unsigned long long bytes = row[2];
  const unsigned long long off0 = (unsigned long long)blockIdx.x * 32768ull;
  if (off0 >= bytes) return;
  const bool vec = ((row[0] | row[1]) & 15ull) == 0;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const unsigned long long off = off0 + ((unsigned long long)it * 256 + threadIdx.x) * 16ull;
    if (off >= bytes) break;
    if (vec && off + 16 <= bytes) {
      *reinterpret_cast<uint4 *>(dst + off) = *reinterpret_cast<const uint4 *>(src + off);
    } else {
      const unsigned long long end = off + 16 < bytes ? off + 16 : bytes;
      for (unsigned long long q = off; q < end; ++q) dst[q] = src[q];
    }
  }
}
}  // namespace

PN2_API int pn2_multi_copy(int n, const void *table, long long max_bytes, void *stream_) {
  if (n <= 0 || max_bytes <= 0) return 0;
  if (!table || n > 65535) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)((max_bytes + 32767) / 32768), n), dim3(256), 0,
                     (hipStream_t)stream_, static_cast<const unsigned long long *>(table));
  return pn2_launch_status();
}

PN2_API const char *pn2_error_string(int code) { return hipGetErrorString((hipError_t)code); }
