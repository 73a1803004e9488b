// tests/loss_host.cpp -- host build of the fused loss' arithmetic (TEST INFRASTRUCTURE).
//
// Compiles 3dioumatch_amd/csrc/loss_core.h -- the very functions the gfx950 kernels call -- with
// the host compiler and runs them from plain loops in the order of csrc/votenet_loss.hip, so that
// the loss values, statistics and gradients can be compared with autograd on a machine without a
// GPU (tests/test_fused_loss.py).  Pointers are HOST pointers here.  Built on demand:
//   g++ -O2 -shared -fPIC -ffp-contract=off -o tests/_loss_host.so tests/loss_host.cpp
#include <vector>

#include "../3dioumatch_amd/csrc/loss_core.h"

extern "C" int host_loss_scratch_floats(const VnLossArgs *args) {
  return args->B * loss_blocks_per_scene(args->K, args->S) * ACC_COUNT;
}

extern "C" int host_loss_decode(const VnLossArgs *args) {
  const LossArgs &a = *args;
  for (int b = 0; b < a.B; ++b) {
    for (int k = 0; k < a.K; ++k) decode_prediction(a, b, k);
    for (int g = 0; g < a.G; ++g) decode_ground_truth(a, b, g);
  }
  return 0;
}

static SceneView scene(const LossArgs &a, int b, std::vector<float> &centers) {
  centers.resize((size_t)a.K * 3);
  for (int k = 0; k < a.K; ++k)
    for (int d = 0; d < 3; ++d) centers[k * 3 + d] = lt_at(a.center, b, k, d);
  SceneView sv;
  sv.gt_center = a.center_label + (long long)b * a.G * 3;
  sv.gt_mask = a.box_label_mask + (long long)b * a.G;
  sv.center = centers.data();
  sv.nearest = a.gt_nearest + (long long)b * a.G;
  return sv;
}

extern "C" int host_loss_forward_backward(const VnLossArgs *args) {
  const LossArgs &a = *args;
  float acc[ACC_COUNT] = {};
  std::vector<float> centers;
  for (int b = 0; b < a.B; ++b) {  // launch "terms"
    const SceneView sv = scene(a, b, centers);
    for (int k = 0; k < a.K; ++k) loss_proposal(a, sv, b, k, acc);
    for (int g = 0; g < a.G; ++g) loss_ground_truth(a, sv, b, g, acc);
    for (int s = 0; s < a.S; ++s) {
      float m;
      loss_seed(a, b, s, acc, &m);
    }
  }
  loss_stats(a, acc);  // launch "finalize"
  for (int b = 0; b < a.B; ++b) {
    const SceneView sv = scene(a, b, centers);
    for (int k = 0; k < a.K; ++k) finalize_proposal(a, sv, b, k, acc);
    for (int s = 0; s < a.S; ++s) {
      float scratch[ACC_COUNT] = {}, m;
      const int arg = loss_seed(a, b, s, scratch, &m);
      vote_grad(a, b, s, arg, m, acc);
    }
  }
  return 0;
}
