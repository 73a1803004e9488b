// 3dioumatch_amd/csrc/votenet_loss.hip -- the supervised VoteNet-IoU loss as three launches (gfx950).
//
// The arithmetic lives in loss_core.h (shared with the host test harness); this file is the
// parallel schedule around it:
//
//   votenet_loss_decode            one lane per proposal / ground-truth slot: boxes for the per-scene
//                                  IoU kernel (iou3d_scene_best_iou3d runs after this launch)
//   votenet_loss_forward_backward  loss_terms_kernel: per scene ceil(K/256) proposal workgroups, one
//                                  ground-truth workgroup and ceil(S/256) seed workgroups.  The
//                                  scene's GT centres / mask (and, for the GT workgroup, the K
//                                  predicted centres) are staged in LDS, so the 64- and K-long
//                                  nearest-neighbour loops never wait on global memory; every lane
//                                  adds its terms to register partial sums and writes its
//                                  unnormalised gradient rows; one row of partial sums per workgroup.
//                                  loss_finalize_kernel (same grid): every workgroup adds up the
//                                  partial rows (a few dozen), then normalises its gradient rows,
//                                  folds in the GT -> nearest-centre term and writes the vote
//                                  gradients; one lane writes the statistics.
//
// (A first version ran everything in ONE 1024-lane workgroup to get the global sums from a single
// LDS reduction: 482 us, every runtime-bound loop paying a global-load latency per iteration on a
// single CU.)
//
// What it replaces: ~150 forward and ~150 backward tensor kernels per train step
// (models/loss_helper_labeled.py:28-370 as mirrored by votenet/losses.py).
#include "common.h"
#define LOSS_HD __host__ __device__ __forceinline__
#include "loss_core.h"

namespace {

constexpr int kMaxG = 256, kMaxK = 2048;

__global__ void __launch_bounds__(256) loss_decode_kernel(LossArgs a) {
  const int item = blockIdx.x * blockDim.x + threadIdx.x;
  const int props = a.B * a.K;
  if (item < props) {
    decode_prediction(a, item / a.K, item % a.K);
  } else if (item < props + a.B * a.G) {
    const int t = item - props;
    decode_ground_truth(a, t / a.G, t % a.G);
  }
}

struct Roles {
  int proposal_blocks, gt_block, per_scene;
};

__device__ __forceinline__ Roles roles(const LossArgs &a) {
  Roles r;
  r.proposal_blocks = (a.K + kLossBlock - 1) / kLossBlock;
  r.gt_block = r.proposal_blocks;
  r.per_scene = loss_blocks_per_scene(a.K, a.S);
  return r;
}

__device__ __forceinline__ void stage_ground_truth(const LossArgs &a, int b, float *gtc, float *gtm) {
  for (int t = threadIdx.x; t < a.G * 3; t += kLossBlock) gtc[t] = a.center_label[(long long)b * a.G * 3 + t];
  for (int t = threadIdx.x; t < a.G; t += kLossBlock) gtm[t] = a.box_label_mask[(long long)b * a.G + t];
}

__global__ void __launch_bounds__(kLossBlock) loss_terms_kernel(LossArgs a) {
  __shared__ float gtc[kMaxG * 3], gtm[kMaxG], ctr[kMaxK * 3];
  __shared__ float partial[kLossBlock / kWave][ACC_COUNT];
  const Roles r = roles(a);
  const int role = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  stage_ground_truth(a, b, gtc, gtm);
  if (role == r.gt_block)
    for (int t = tid; t < a.K * 3; t += kLossBlock) ctr[t] = lt_at(a.center, b, t / 3, t % 3);
  __syncthreads();
  SceneView sv;
  sv.gt_center = gtc; sv.gt_mask = gtm; sv.center = ctr; sv.nearest = nullptr;
  float acc[ACC_COUNT];
#pragma unroll
  for (int i = 0; i < ACC_COUNT; ++i) acc[i] = 0.0f;
  if (role < r.proposal_blocks) {
    const int k = role * kLossBlock + tid;
    if (k < a.K) loss_proposal(a, sv, b, k, acc);
  } else if (role == r.gt_block) {
    for (int g = tid; g < a.G; g += kLossBlock) loss_ground_truth(a, sv, b, g, acc);
  } else {
    const int s = (role - r.gt_block - 1) * kLossBlock + tid;
    float m;
    if (s < a.S) loss_seed(a, b, s, acc, &m);
  }
  // one row of partial sums per workgroup: wave shuffle, then the four wave results through LDS
  const int lane = lane_id(), wave = tid / kWave;
#pragma unroll
  for (int i = 0; i < ACC_COUNT; ++i) {
    float v = acc[i];
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    if (lane == 0) partial[wave][i] = v;
  }
  __syncthreads();
  if (tid < ACC_COUNT) {
    float v = 0.0f;
    for (int w = 0; w < kLossBlock / kWave; ++w) v += partial[w][tid];
    a.partials[((long long)b * r.per_scene + role) * ACC_COUNT + tid] = v;
  }
}

__global__ void __launch_bounds__(kLossBlock) loss_finalize_kernel(LossArgs a) {
  __shared__ float gtc[kMaxG * 3], gtm[kMaxG];
  __shared__ int nearest[kMaxG];
  __shared__ float total[ACC_COUNT];
  __shared__ float share[kLossBlock / 32][32];
  const Roles r = roles(a);
  const int role = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  {  // the global sums: lane (group, column) adds every 8th partial row, then the 8 shares
    const int col = tid & 31, group = tid >> 5, rows = a.B * r.per_scene;
    float v = 0.0f;
    if (col < ACC_COUNT)
      for (int row = group; row < rows; row += kLossBlock / 32)
        v += a.partials[(long long)row * ACC_COUNT + col];
    share[group][col] = v;
    __syncthreads();
    if (tid < ACC_COUNT) {
      float t = 0.0f;
      for (int g = 0; g < kLossBlock / 32; ++g) t += share[g][tid];
      total[tid] = t;
    }
  }
  if (role < r.proposal_blocks) {
    stage_ground_truth(a, b, gtc, gtm);
    for (int t = tid; t < a.G; t += kLossBlock) nearest[t] = a.gt_nearest[(long long)b * a.G + t];
  }
  __syncthreads();
  float acc[ACC_COUNT];
#pragma unroll
  for (int i = 0; i < ACC_COUNT; ++i) acc[i] = total[i];
  if (role < r.proposal_blocks) {
    SceneView sv;
    sv.gt_center = gtc; sv.gt_mask = gtm; sv.center = nullptr; sv.nearest = nearest;
    const int k = role * kLossBlock + tid;
    if (k < a.K) finalize_proposal(a, sv, b, k, acc);
  } else if (role == r.gt_block) {
    if (b == 0 && tid == 0) loss_stats(a, acc);
  } else {
    const int s = (role - r.gt_block - 1) * kLossBlock + tid;
    if (s < a.S) {
      float scratch[ACC_COUNT] = {}, m;
      const int arg = loss_seed(a, b, s, scratch, &m);
      vote_grad(a, b, s, arg, m, acc);
    }
  }
}

bool valid(const VnLossArgs *a) {
  if (a && a->consistency && (a->S != 0 || a->VF != 0)) return false;  // no vote term there
  return a && a->B > 0 && a->K > 0 && a->G > 0 && (a->consistency || (a->S > 0 && a->VF > 0)) && a->NH > 0 && a->NS > 0 &&
         a->NC > 0 && (a->NI == 1 || a->NI == a->NC) && a->G <= kMaxG && a->K <= kMaxK &&
         a->B <= 65535;
}

}  // namespace

extern "C" __attribute__((visibility("default")))
int votenet_loss_scratch_floats(const VnLossArgs *args) {
  if (!valid(args)) return 0;
  return args->B * loss_blocks_per_scene(args->K, args->S) * ACC_COUNT;
}

extern "C" __attribute__((visibility("default")))
int votenet_loss_decode(const VnLossArgs *args, void *stream) {
  if (!valid(args)) return (int)hipErrorInvalidValue;
  const int items = args->B * (args->K + args->G);
  hipLaunchKernelGGL(loss_decode_kernel, dim3((items + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, *args);
  return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default")))
int votenet_loss_forward_backward(const VnLossArgs *args, void *stream) {
  if (!valid(args) || !args->partials) return (int)hipErrorInvalidValue;
  const dim3 grid(loss_blocks_per_scene(args->K, args->S), args->B);
  hipLaunchKernelGGL(loss_terms_kernel, grid, dim3(kLossBlock), 0, (hipStream_t)stream, *args);
  hipLaunchKernelGGL(loss_finalize_kernel, grid, dim3(kLossBlock), 0, (hipStream_t)stream, *args);
  return (int)hipGetLastError();
}


// ---- decode_scores (models/proposal_module.py:24-54) and its backward -----------------------------
// A lane per proposal: its column of the head output (b, c, k) -- coalesced along k -- in, the rows
// of the nine named predictions out.
namespace {

struct DecodeArgs {
  int b, k, nh, ns, nc;
  const float *net, *agg_xyz, *mean_size;
  float *objectness, *center, *heading_scores, *heading_resn, *heading_res, *size_scores, *size_resn,
      *size_res, *sem_cls;
};

__device__ __forceinline__ float softplus1(float x) {  // F.softplus: beta 1, threshold 20
  return x > 20.0f ? x : log1pf(expf(x));
}

// One lane per (cloud, channel, proposal) -- the element order of `net` itself, so the read is one
// coalesced load per lane and nothing is a loop: with one lane per PROPOSAL (rounds 5: 8 workgroups
// on 256 CUs, each lane a chain of ~97 dependent strided loads + expf / log1pf) the two kernels took
// 55 + 60 us for 0.8 MB.
__global__ void __launch_bounds__(256) decode_scores_kernel(DecodeArgs a) {
  const int c_total = 5 + 2 * a.nh + 4 * a.ns + a.nc;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)a.b * c_total * a.k) return;
  const int ki = (int)(t % a.k);
  const long long bc = t / a.k;
  const int c = (int)(bc % c_total), bi = (int)(bc / c_total);
  const long long i = (long long)bi * a.k + ki;
  const float v = a.net[t];
  int j = c;
  if (j < 2) { a.objectness[i * 2 + j] = v; return; }
  j -= 2;
  if (j < 3) { a.center[i * 3 + j] = a.agg_xyz[i * 3 + j] + v; return; }
  j -= 3;
  if (j < a.nh) { a.heading_scores[i * a.nh + j] = v; return; }
  j -= a.nh;
  if (j < a.nh) {
    const float per = (float)(M_PI / (double)a.nh);   // np.pi / nh as a Python float, then fp32
    a.heading_resn[i * a.nh + j] = v;
    a.heading_res[i * a.nh + j] = v * per;
    return;
  }
  j -= a.nh;
  if (j < a.ns) { a.size_scores[i * a.ns + j] = v; return; }
  j -= a.ns;
  if (j < a.ns * 3) {
    const float u = softplus1(v) - 1.0f;
    a.size_resn[i * a.ns * 3 + j] = u;
    a.size_res[i * a.ns * 3 + j] = u * a.mean_size[j];
    return;
  }
  j -= a.ns * 3;
  a.sem_cls[i * a.nc + j] = v;
}

struct DecodeGradArgs {
  int b, k, nh, ns, nc;
  const float *net, *mean_size;
  const float *g_objectness, *g_center, *g_heading_scores, *g_heading_resn, *g_heading_res,
      *g_size_scores, *g_size_resn, *g_size_res, *g_sem_cls;
  float *d_net;
};

__global__ void __launch_bounds__(256) decode_scores_grad_kernel(DecodeGradArgs a) {
  const int c_total = 5 + 2 * a.nh + 4 * a.ns + a.nc;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)a.b * c_total * a.k) return;
  const int ki = (int)(t % a.k);
  const long long bc = t / a.k;
  const int c = (int)(bc % c_total), bi = (int)(bc / c_total);
  const long long i = (long long)bi * a.k + ki;
  auto get = [&](const float *g, long long at) { return g ? g[at] : 0.0f; };
  int j = c;
  float out;
  if (j < 2) {
    out = get(a.g_objectness, i * 2 + j);
  } else if ((j -= 2) < 3) {
    out = get(a.g_center, i * 3 + j);
  } else if ((j -= 3) < a.nh) {
    out = get(a.g_heading_scores, i * a.nh + j);
  } else if ((j -= a.nh) < a.nh) {
    const float per = (float)(M_PI / (double)a.nh);
    out = get(a.g_heading_resn, i * a.nh + j) + get(a.g_heading_res, i * a.nh + j) * per;
  } else if ((j -= a.nh) < a.ns) {
    out = get(a.g_size_scores, i * a.ns + j);
  } else if ((j -= a.ns) < a.ns * 3) {
    const float x = a.net[t];
    const float slope = x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));   // softplus' = sigmoid
    const float g = get(a.g_size_resn, i * a.ns * 3 + j) + get(a.g_size_res, i * a.ns * 3 + j) * a.mean_size[j];
    out = g * slope;
  } else {
    j -= a.ns * 3;
    out = get(a.g_sem_cls, i * a.nc + j);
  }
  a.d_net[t] = out;
}

}  // namespace

extern "C" __attribute__((visibility("default")))
int votenet_decode_scores(int b, int k, int nh, int ns, int nc, const float *net, const float *agg_xyz,
                          const float *mean_size, float *objectness, float *center,
                          float *heading_scores, float *heading_resn, float *heading_res,
                          float *size_scores, float *size_resn, float *size_res, float *sem_cls,
                          void *stream) {
  if (b <= 0 || k <= 0) return 0;
  if (nh <= 0 || ns <= 0 || nc <= 0) return (int)hipErrorInvalidValue;
  const DecodeArgs a = {b, k, nh, ns, nc, net, agg_xyz, mean_size, objectness, center, heading_scores,
                        heading_resn, heading_res, size_scores, size_resn, size_res, sem_cls};
  const long long total = (long long)b * (5 + 2 * nh + 4 * ns + nc) * k;
  hipLaunchKernelGGL(decode_scores_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default")))
int votenet_decode_scores_grad(int b, int k, int nh, int ns, int nc, const float *net,
                               const float *mean_size, const float *g_objectness, const float *g_center,
                               const float *g_heading_scores, const float *g_heading_resn,
                               const float *g_heading_res, const float *g_size_scores,
                               const float *g_size_resn, const float *g_size_res, const float *g_sem_cls,
                               float *d_net, void *stream) {
  if (b <= 0 || k <= 0) return 0;
  if (nh <= 0 || ns <= 0 || nc <= 0) return (int)hipErrorInvalidValue;
  const DecodeGradArgs a = {b, k, nh, ns, nc, net, mean_size, g_objectness, g_center, g_heading_scores,
                            g_heading_resn, g_heading_res, g_size_scores, g_size_resn, g_size_res,
                            g_sem_cls, d_net};
  const long long total = (long long)b * (5 + 2 * nh + 4 * ns + nc) * k;
  hipLaunchKernelGGL(decode_scores_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                     0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
