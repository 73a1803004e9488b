/*
 * include/iou3d_hip.h -- C ABI of the MI355X-native rotated-box IoU / 3-D NMS operators.
 *
 * Drop-in boundary for `pcdet.ops.iou3d_nms.iou3d_nms_cuda` (reference pybind surface:
 * OpenPCDet/pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17).  Each entry point replaces
 * one of the reference's launchers (declared at iou3d_nms.cpp:43-46), with an explicit
 * hipStream_t (void*) -- the reference launches on the legacy default stream
 * (iou3d_nms_kernel.cu:396,407,418,426) -- and an `int` hipError_t return instead of exit().
 *
 * Boxes are rows of 7 float32: (x, y, z, dx, dy, dz, heading); matrices are row-major.
 * Pointers are DEVICE pointers unless the name ends in _host.
 */
#ifndef IOU3D_HIP_H
#define IOU3D_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* replaces boxesoverlapLauncher (iou3d_nms.cpp:43, iou3d_nms_kernel.cu:249-262,389-399):
 * ans[i,j] = BEV intersection area of boxes_a[i] and boxes_b[j]. */
int iou3d_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                            float *ans_overlap, void *stream);

/* replaces boxesioubevLauncher (iou3d_nms.cpp:44, iou3d_nms_kernel.cu:264-278,400-410):
 * ans[i,j] = overlap / max(area_a + area_b - overlap, 1e-8). */
int iou3d_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                        float *ans_iou, void *stream);

/* replaces nmsLauncher (iou3d_nms.cpp:45, iou3d_nms_kernel.cu:280-324,413-419): suppression
 * bit mask (boxes_num, ceil(boxes_num/64)) u64 using the fork's 3-D IoU (iou_bev_3D,
 * iou3d_nms_kernel.cu:237-247).  boxes must be sorted by score, descending. */
int iou3d_nms_mask(const float *boxes, unsigned long long *mask, int boxes_num,
                   float nms_overlap_thresh, void *stream);

/* replaces nmsNormalLauncher (iou3d_nms.cpp:46, iou3d_nms_kernel.cu:341-385,422-427):
 * same tiling with the axis-aligned BEV IoU (iou_normal, :327-338). */
int iou3d_nms_normal_mask(const float *boxes, unsigned long long *mask, int boxes_num,
                          float nms_overlap_thresh, void *stream);

/* replaces the whole of nms_gpu / nms_normal_gpu (iou3d_nms.cpp:90-138,141-190): mask kernel
 * plus the greedy scan, which runs ON THE DEVICE here (one wavefront; the reference copies
 * the mask to the host and scans there).  keep_dev receives the kept indices (int64, the
 * upstream-OpenPCDet contract -- the fork's C++ misreads the LongTensor as int32,
 * iou3d_nms.cpp:98); num_out_dev receives the count (int32).  mask_ws must hold
 * boxes_num*ceil(boxes_num/64) u64.  normal != 0 selects iou_normal. */
int iou3d_nms(const float *boxes, int boxes_num, float nms_overlap_thresh, int normal,
              unsigned long long *mask_ws, long long *keep_dev, int *num_out_dev, void *stream);

/* replaces boxes_iou_bev_cpu (iou3d_cpu.cpp:232-252), the reference's only native CPU op on
 * this path: HOST pointers, single-threaded, same arithmetic as iou3d_boxes_iou_bev. */
int iou3d_boxes_iou_bev_cpu(int num_a, const float *boxes_a_host, int num_b,
                            const float *boxes_b_host, float *ans_iou_host);

/* ---- addition (the reference composes this in Python, iou3d_nms_utils.py:48-81) ---- */

/* replaces the Python composition boxes_iou3d_gpu (iou3d_nms_utils.py:48-81): 3-D IoU matrix in
 * one kernel: BEV overlap x z-overlap / clamp(vol_a + vol_b - ov3d, 1e-6). */
int iou3d_boxes_iou3d(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                      float *ans_iou3d, void *stream);

/* replaces boxes_iou3d_gpu over ALL (scene, scene') pairs followed by the block-diagonal
 * max / gather of compute_iou_labels (loss_helper_iou.py:98-111): boxes_a (scenes, num_a, 7),
 * boxes_b (scenes, num_b, 7) -> best_iou (scenes, num_a) f32 and best_idx (scenes, num_a) i32 =
 * the first box of the SAME scene with the largest 3-D IoU (index 0 when every IoU is 0). */
int iou3d_scene_best_iou3d(int scenes, int num_a, const float *boxes_a, int num_b,
                           const float *boxes_b, float *best_iou, int *best_idx, void *stream);

/* ---- evaluation path ---- */

/* replaces get_iou_obb = box3d_iou(...)[0] (utils/eval_det.py:74-77, utils/box_util.py:112-137)
 * for all pairs: a (n,8,3), b (m,8,3) float32 corners in the upright camera frame (vertex order
 * of get_3d_box, utils/box_util.py:335-358) -> iou (n,m) float64.  Footprint clipping follows
 * polygon_clip (box_util.py:23-69) in float64 source order; the clipped polygon's area is its
 * shoelace sum (the reference asks scipy's ConvexHull, box_util.py:77-88; equal to rounding), 0
 * for fewer than 3 vertices (the reference raises QhullError there). */
int iou3d_corners_iou3d(int n, const float *a, int m, const float *b, double *iou, void *stream);

/* replaces the per-detection loop of eval_det_cls (utils/eval_det.py:128-141): for detection d
 * the ground-truth boxes gt[gt_begin[d] .. gt_begin[d]+gt_count[d]) (its scan, its class) ->
 * ovmax[d] f64 = largest IoU, jmax[d] i32 = first box reaching it (strict `>` update); -inf / -1
 * when gt_count[d] == 0.  det (nd,8,3), gt (ng,8,3) float32 corners. */
int iou3d_corners_best_match(int nd, const float *det, const int *gt_begin, const int *gt_count,
                             const float *gt, double *ovmax, int *jmax, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IOU3D_HIP_H */
