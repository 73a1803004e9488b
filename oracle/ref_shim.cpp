// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Builds the REFERENCE's own CPU source for the iou3d path from where it lies
// (OpenPCDet/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp, pulled in through the -I path the
// Makefile passes; nothing is copied into this repository) and exposes a plain C ABI
// around it so tests can pin oracle/iou3d_oracle.c against the real thing:
//   ref_boxes_iou_bev_cpu  -> the reference entry point boxes_iou_bev_cpu
//                             (iou3d_cpu.cpp:232-252), called with at::Tensor views
//   ref_box_overlap_matrix -> the reference's internal box_overlap
//                             (iou3d_cpu.cpp:128-220) evaluated per pair
// The output (.so) goes to oracle/_ref/, which is git-ignored but travels to the GPU
// box.  This file only compiles where /root/reference exists.
#include <iou3d_cpu.cpp>  // the reference translation unit itself

extern "C" {

int ref_boxes_iou_bev_cpu(int na, const float *boxes_a, int nb, const float *boxes_b,
                          float *ans) {
  auto opt = at::TensorOptions().dtype(at::kFloat).device(at::kCPU);
  at::Tensor ta = at::from_blob(const_cast<float *>(boxes_a), {na, 7}, opt);
  at::Tensor tb = at::from_blob(const_cast<float *>(boxes_b), {nb, 7}, opt);
  at::Tensor to = at::from_blob(ans, {na, nb}, opt);
  return boxes_iou_bev_cpu(ta, tb, to);
}

void ref_box_overlap_matrix(int na, const float *boxes_a, int nb, const float *boxes_b,
                            float *ans) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      ans[(size_t)i * nb + j] = box_overlap(boxes_a + i * 7, boxes_b + j * 7);
}

}  // extern "C"
