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
