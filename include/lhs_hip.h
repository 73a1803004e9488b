/* include/lhs_hip.h -- C ABI of the pseudo-label filter's NMS (semi-supervised step, SURVEY
 * section 8(f) rank 1).  Plain pointers to DEVICE memory, explicit stream, returns hipError_t. */
#ifndef LHS_HIP_H
#define LHS_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* replaces the host-side loop of get_pseudo_labels (models/loss_helper_unlabeled.py:447-487):
 * get_3d_box per box (utils/box_util.py:335-358) -> axis-aligned bounds in the camera frame ->
 * lhs_3d_faster_samecls (utils/nms.py:168-214), per scene.  center (scenes,n,3) f32, size
 * (scenes,n,3) f64, heading (scenes,n) f64, score (scenes,n) f32, cls (scenes,n) i64, n <= 64;
 * picked (scenes,n) i32 = 1 for every index the reference's function returns. */
int lhs_nms_samecls(int scenes, int n, const float *center, const double *size,
                    const double *heading, const float *score, const long long *cls, double thresh,
                    int old_type, int *picked, void *stream);

/* replaces the per-scene numpy NMS of the evaluation path (models/ap_helper.py:170-203):
 * nms_3d_faster (same_class = 0, utils/nms.py:77-116) / nms_3d_faster_samecls (same_class = 1,
 * utils/nms.py:118-166) on the axis-aligned camera-frame bounds of get_3d_box
 * (utils/box_util.py:335-358).  Arguments as lhs_nms_samecls; n <= 1024; picked (scenes,n) i32. */
int lhs_nms3d_aabb(int scenes, int n, const float *center, const double *size,
                   const double *heading, const float *score, const long long *cls, double thresh,
                   int old_type, int same_class, int *picked, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LHS_HIP_H */
