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

/* The rest of the pseudo-label filter around that NMS: get_pseudo_labels and the label transforms
 * of get_unlabeled_loss (models/loss_helper_unlabeled.py:364-445, :489-538, trans_center :24-36,
 * trans_size :39-51) as two launches (select, then -- after lhs_nms_samecls on the select's boxes --
 * finish) instead of ~55 tensor kernels.  All pointers: DEVICE memory, contiguous.  S unlabeled
 * scenes, K teacher proposals each (K <= 1024), n = 64 label slots. */
typedef struct LhsPseudoArgs {
  int S, K, NC, NI, NH, NS;            /* NI: 1 or NC IoU channels */
  float obj_threshold, cls_threshold, iou_threshold;
  int use_nms;                         /* finish: also require picked != 0 */
  /* teacher outputs of the unlabeled scenes */
  const float *objectness;             /* (S,K,2)   */
  const float *sem_cls;                /* (S,K,NC)  */
  const float *iou;                    /* (S,K,NI)  */
  const float *heading_scores;         /* (S,K,NH)  */
  const float *heading_residuals;      /* (S,K,NH)  */
  const float *size_scores;            /* (S,K,NS)  */
  const float *size_residuals;         /* (S,K,NS,3) */
  const float *center;                 /* (S,K,3)   */
  const float *vote_xyz;               /* (S,K,3) aggregated votes */
  const float *mean_size;              /* (NS,3)    */
  /* the student's augmentation of each scene */
  const long long *flip_x, *flip_y;    /* (S)       */
  const float *rot_mat;                /* (S,3,3)   */
  const float *scale;                  /* (S,3)     */
  /* select -> (NMS) -> finish: the 64 best survivors per scene, in score order */
  float *box_center;                   /* (S,64,3) teacher frame              */
  double *box_size;                    /* (S,64,3) float64 decode for the NMS */
  double *box_heading;                 /* (S,64)                              */
  float *box_score;                    /* (S,64) objectness * predicted IoU   */
  int *passed, *negative;              /* (S,64) thresholds passed / objectness < 0.1 */
  float *false_xyz;                    /* (S,64,3) votes of the slots         */
  const int *picked;                   /* (S,64) of lhs_nms_samecls (finish; NULL when !use_nms) */
  /* labels (finish writes label_mask and the transformed centres / size residuals; select the rest) */
  long long *label_mask;               /* (S,64) */
  float *center_label;                 /* (S,64,3) student frame, -1000-based where label_mask = 0 */
  float *false_center_label;           /* (S,64,3) */
  long long *sem_cls_label, *heading_label, *size_label;   /* (S,64) */
  float *heading_residual_label;       /* (S,64) */
  float *size_residual_label;          /* (S,64,3): select the teacher's, finish rescales in place */
  float *iou_label;                    /* (S,64) */
  float *pseudo_gt_ratio;              /* 1: share of slots that passed the thresholds (before NMS) */
} LhsPseudoArgs;
/* scores, threshold masks, the 64 best per scene in score order, their decoded boxes
 * (models/loss_helper_unlabeled.py:364-445); `args`: a HOST struct of DEVICE pointers */
int lhs_pseudo_select(const LhsPseudoArgs *args, void *stream);
/* NMS verdict, -1000 placeholders, labels in the student's frame
 * (models/loss_helper_unlabeled.py:489-538 with trans_center :24-36 and trans_size :39-51) */
int lhs_pseudo_finish(const LhsPseudoArgs *args, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LHS_HIP_H */
