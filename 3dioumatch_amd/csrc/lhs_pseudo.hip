// 3dioumatch_amd/csrc/lhs_pseudo.hip -- the pseudo-label filter of the semi-supervised step around
// its NMS kernel (lhs_nms.hip), gfx950.
//
// What it replaces: get_pseudo_labels (models/loss_helper_unlabeled.py:364-538, mirrored with
// tensor operations in votenet/losses_unlabeled.py) -- two softmaxes, a sigmoid, three threshold
// masks, a stable descending sort of the 256 scores of a scene, fifteen gathers at the sorted
// indices, the float64 box decoding for the NMS -- and, after the NMS, the masks and the transforms
// of the labels into the student's augmented frame (trans_center :24-36, trans_size :39-51): ~55
// kernels of a few hundred elements each, 5-6 us apiece in a 11 ms step.
//
//   lhs_pseudo_select  one workgroup per scene, a lane per proposal: scores and masks in
//                      registers, the sort as a RANK (how many proposals have a larger key, or the
//                      same key and a smaller index: what a stable descending sort computes), and
//                      the lane whose rank is below 64 writes its slot directly -- no index array,
//                      no gathers.
//   lhs_pseudo_finish  a lane per slot: NMS verdict, -1000 placeholders, flips / rotation / scale.
//
// Arithmetic follows the tensor version operation by operation in fp32 (softmax as
// exp(x - max) / sum, sigmoid as 1 / (1 + exp(-x)), key = (objectness * class probability) *
// mask); discrete outputs can differ from it only where two keys or a threshold are one ulp apart.
#include "common.h"
#include "../../include/lhs_hip.h"

namespace {

constexpr int kSlots = 64;      // MAX_NUM_OBJ
constexpr int kMaxK = 1024;

__device__ __forceinline__ int first_max(const float *row, int n) {
  int best = 0;
  float m = row[0];
  for (int j = 1; j < n; ++j)
    if (row[j] > m) { m = row[j]; best = j; }
  return best;
}

__global__ void __launch_bounds__(256) pseudo_select_kernel(LhsPseudoArgs a) {
  __shared__ float key[kMaxK];
  const int s = blockIdx.x, tid = threadIdx.x;
  // pass 1: the key of every proposal (kept in LDS for the ranking)
  for (int k = tid; k < a.K; k += 256) {
    const long long sk = (long long)s * a.K + k;
    const float s0 = a.objectness[sk * 2], s1 = a.objectness[sk * 2 + 1];
    const float m = s0 > s1 ? s0 : s1;
    const float e0 = expf(s0 - m), e1 = expf(s1 - m);
    const float pos = e1 / (e0 + e1);
    const float *sem = a.sem_cls + sk * a.NC;
    const int cls = first_max(sem, a.NC);
    float sum = 0.0f;
    for (int j = 0; j < a.NC; ++j) sum += expf(sem[j] - sem[cls]);
    const float max_cls = 1.0f / sum;
    const float x = a.iou[sk * a.NI + (a.NI > 1 ? cls : 0)];
    const float iou = 1.0f / (1.0f + expf(-x));
    const bool ok = max_cls > a.cls_threshold && pos > a.obj_threshold && iou > a.iou_threshold;
    // (a NaN / Inf logit makes the key NaN: every comparison of the ranking below is then false,
    //  several proposals take rank 0 and other slots are never written -- the tensor path's
    //  argsort always yields a permutation.  Such a proposal is not a pseudo label: key 0.)
    const float v = pos * max_cls;
    key[k] = (ok && isfinite(v)) ? v : 0.0f;
  }
  __syncthreads();
  // pass 2: rank, and the slots
  for (int k = tid; k < a.K; k += 256) {
    const float mine = key[k];
    int rank = 0;
    for (int j = 0; j < a.K; ++j) {
      const float o = key[j];
      rank += (o > mine || (o == mine && j < k)) ? 1 : 0;
    }
    if (rank >= kSlots) continue;
    const long long sk = (long long)s * a.K + k, slot = (long long)s * kSlots + rank;
    // (recomputed: cheaper than keeping six values per proposal in LDS)
    const float s0 = a.objectness[sk * 2], s1 = a.objectness[sk * 2 + 1];
    const float m = s0 > s1 ? s0 : s1;
    const float e0 = expf(s0 - m), e1 = expf(s1 - m);
    const float pos = e1 / (e0 + e1), neg = e0 / (e0 + e1);
    const float *sem = a.sem_cls + sk * a.NC;
    const int cls = first_max(sem, a.NC);
    const float x = a.iou[sk * a.NI + (a.NI > 1 ? cls : 0)];
    const float iou = 1.0f / (1.0f + expf(-x));
    const bool ok = mine > 0.0f;   // key = pos * max_cls * mask, pos and max_cls > 0
    const int hl = first_max(a.heading_scores + sk * a.NH, a.NH);
    const int sl = first_max(a.size_scores + sk * a.NS, a.NS);
    const float h_res = a.heading_residuals[sk * a.NH + hl];
    a.passed[slot] = ok ? 1 : 0;
    a.negative[slot] = neg > 0.9f ? 1 : 0;
    a.sem_cls_label[slot] = cls;
    a.heading_label[slot] = hl;
    a.size_label[slot] = sl;
    a.heading_residual_label[slot] = h_res;
    a.iou_label[slot] = iou;
    a.box_score[slot] = pos * iou;
    for (int d = 0; d < 3; ++d) {
      const float res = a.size_residuals[(sk * a.NS + sl) * 3 + d];
      a.size_residual_label[slot * 3 + d] = res;
      a.box_size[slot * 3 + d] = (double)a.mean_size[sl * 3 + d] + (double)res;   // numpy decode: float64
      a.box_center[slot * 3 + d] = a.center[sk * 3 + d];
      a.false_xyz[slot * 3 + d] = a.vote_xyz[sk * 3 + d];
    }
    double angle = 0.0;   // config.class2angle_f64
    if (a.NH > 1) {
      angle = (double)hl * (2.0 * M_PI / (double)a.NH) + (double)h_res;
      if (angle > M_PI) angle -= 2.0 * M_PI;
    }
    a.box_heading[slot] = angle;
  }
}

__global__ void __launch_bounds__(kSlots) pseudo_finish_kernel(LhsPseudoArgs a) {
  const int s = blockIdx.x, t = threadIdx.x;
  const long long slot = (long long)s * kSlots + t;
  const bool keep = a.passed[slot] != 0 && (!a.use_nms || a.picked[slot] != 0);
  a.label_mask[slot] = keep ? 1 : 0;
  const bool fx = a.flip_x[s] != 0, fy = a.flip_y[s] != 0;
  const float *R = a.rot_mat + (long long)s * 9;
  const float *sc = a.scale + (long long)s * 3;
  auto to_student = [&](float x, float y, float z, float *out) {  // trans_center
    x = fx ? -x : x;
    y = fy ? -y : y;
    for (int j = 0; j < 3; ++j) out[j] = ((x * R[j * 3] + y * R[j * 3 + 1]) + z * R[j * 3 + 2]) * sc[j];
  };
  const bool neg = a.negative[slot] != 0;
  float c[3], f[3];
  for (int d = 0; d < 3; ++d) {
    c[d] = keep ? a.box_center[slot * 3 + d] : -1000.0f;
    f[d] = neg ? a.false_xyz[slot * 3 + d] : -1000.0f;
  }
  to_student(c[0], c[1], c[2], a.center_label + slot * 3);
  to_student(f[0], f[1], f[2], a.false_center_label + slot * 3);
  const int sl = (int)a.size_label[slot];
  for (int d = 0; d < 3; ++d) {  // trans_size
    const float base = a.mean_size[sl * 3 + d];
    a.size_residual_label[slot * 3 + d] = (base + a.size_residual_label[slot * 3 + d]) * sc[d] - base;
  }
  if (s == 0) {   // pseudo_gt_ratio: mean of the threshold mask over every slot of every scene
    int n = 0;
    for (int q = 0; q < a.S; ++q) n += __popcll(__ballot(a.passed[(long long)q * kSlots + t] != 0));
    if (t == 0) *a.pseudo_gt_ratio = (float)n / (float)((long long)a.S * kSlots);
  }
}

bool valid(const LhsPseudoArgs *a) {
  return a && a->S > 0 && a->S <= 65535 && a->K >= kSlots && a->K <= kMaxK && a->NC > 0 &&
         (a->NI == 1 || a->NI == a->NC) && a->NH > 0 && a->NS > 0;
}

}  // namespace

extern "C" __attribute__((visibility("default")))
int lhs_pseudo_select(const LhsPseudoArgs *args, void *stream) {
  if (!valid(args)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(pseudo_select_kernel, dim3(args->S), dim3(256), 0, (hipStream_t)stream, *args);
  return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default")))
int lhs_pseudo_finish(const LhsPseudoArgs *args, void *stream) {
  if (!valid(args) || (args->use_nms && !args->picked)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(pseudo_finish_kernel, dim3(args->S), dim3(kSlots), 0, (hipStream_t)stream, *args);
  return (int)hipGetLastError();
}
