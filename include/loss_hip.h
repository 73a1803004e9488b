/* include/loss_hip.h -- C ABI of the fused supervised VoteNet-IoU loss (forward value, statistics
 * and the gradient with respect to every head output), gfx950.  Plain pointers to DEVICE memory,
 * explicit stream, returns hipError_t.  Not in the reference: its loss is ~300 small tensor kernels
 * per step launched from Python (models/loss_helper_labeled.py:28-370, models/loss_helper_iou.py:
 * 52-112, utils/nn_distance.py:16-62). */
#ifndef LOSS_HIP_H
#define LOSS_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* a (B, K, C[, 3]) float tensor with element strides (head outputs are transposed slices of the
 * heads' (B, C, K) outputs; nothing is copied for the kernel) */
typedef struct VnLossTensor {
  const float *p;
  long long sb, sk, sc, sd;
} VnLossTensor;

/* indices into VnLossArgs.stats */
enum {
  VN_ST_LOSS, VN_ST_VOTE, VN_ST_OBJ, VN_ST_CENTER, VN_ST_HCLS, VN_ST_HREG, VN_ST_SCLS, VN_ST_SREG,
  VN_ST_SEM, VN_ST_BOX, VN_ST_IOU, VN_ST_JIT, VN_ST_POS_RATIO, VN_ST_NEG_RATIO, VN_ST_OBJ_ACC,
  VN_ST_OBJ_COUNT, VN_ST_CLS_ACC, VN_ST_PRED_IOU, VN_ST_PRED_IOU_OBJ, VN_ST_IOU_ACC,
  VN_ST_IOU_ACC_OBJ, VN_ST_JIT_ACC, VN_ST_JIT_ACC_OBJ, VN_ST_COUNT
};

typedef struct VnLossArgs {
  int B, K, G, S, VF, N;   /* supervised scenes (the first B of the batch), proposals, GT slots,
                              seeds, votes per seed, points per scene */
  int NH, NS, NC, NI;      /* heading bins, size clusters, classes, IoU channels (1 or NC) */
  int has_jitter;
  int consistency;         /* 1: the consistency loss of the semi-supervised stage on pseudo labels
                              (models/loss_helper_unlabeled.py:137-361): the labels are the pseudo
                              boxes (centres of empty slots already at -1000), there is no vote term
                              (S = VF = 0, vote / seed arguments unused) and no IoU term (iou
                              arguments unused, g_iou* not written), the objectness term is a
                              statistic only (g_obj = 0): loss = 10 (box + 0.1 sem_cls) */
  float grad_scale;        /* every gradient output is d(grad_scale * stats[VN_ST_LOSS]) / d(prediction):
                              1 for the loss alone; w where the caller's objective adds this loss
                              with weight w (train.py:333: detection_loss + 2.0 * unlabeled_loss)
                              and wants ONE gradient buffer for the sum */
  /* labels (loss_helper_labeled.py end_points keys): contiguous, first B scenes are read */
  const float *center_label;             /* (.,G,3) */
  const float *box_label_mask;           /* (.,G)   */
  const long long *heading_class_label;  /* (.,G)   */
  const float *heading_residual_label;   /* (.,G)   */
  const long long *size_class_label;     /* (.,G)   */
  const float *size_residual_label;      /* (.,G,3) */
  const long long *sem_cls_label;        /* (.,G)   */
  const float *vote_label;               /* (.,N,9) */
  const long long *vote_label_mask;      /* (.,N)   */
  const int *seed_inds;                  /* (.,S), row stride seed_inds_stride */
  long long seed_inds_stride;
  const float *mean_size;                /* (NS,3)  */
  /* predictions */
  VnLossTensor agg_xyz, obj, center, h_scores, h_resn, s_scores, s_resn, sem, iou, iou_jit;
  VnLossTensor seed_xyz, vote_xyz;               /* (B,S,3), (B,S*VF,3) */
  VnLossTensor jit_center, jit_size, jit_heading; /* read by votenet_loss_decode only */
  /* boxes: written by votenet_loss_decode, read by iou3d_scene_best_iou3d; its outputs */
  float *boxes;            /* (B, K or 2K, 7): decoded predictions, then the jittered boxes */
  float *gt_boxes;         /* (B, G, 7) */
  const float *iou_lab;    /* (B, K or 2K) */
  const int *iou_assign;   /* (B, K or 2K) */
  /* outputs of votenet_loss_forward_backward */
  float *stats;                  /* VN_ST_COUNT */
  long long *objectness_label;   /* (B,K) */
  float *objectness_mask;        /* (B,K) */
  long long *object_assignment;  /* (B,K) */
  float *g_obj, *g_center, *g_h_scores, *g_h_resn, *g_s_scores, *g_s_resn, *g_sem, *g_iou,
      *g_iou_jit, *g_vote;       /* d(loss)/d(prediction), contiguous (B,K,C[,3]) / (B,S*VF,3) */
  int *gt_nearest;               /* scratch (B,G) */
  float *partials;               /* scratch: votenet_loss_scratch_floats(args) floats */
} VnLossArgs;

/* replaces the box decoding of compute_iou_labels (models/loss_helper_iou.py:60-96) and of the
 * ground truth (:79-88): boxes / gt_boxes <- [centre, size, -heading]; `args` is a HOST struct of
 * DEVICE pointers. */
int votenet_loss_decode(const VnLossArgs *args, void *stream);

/* replaces get_labeled_loss (models/loss_helper_labeled.py:300-370) after the IoU labels are
 * known: every loss term, every statistic the training loop logs, the objectness labels / mask /
 * assignment, and the gradient of `stats[VN_ST_LOSS]` with respect to each prediction.  Two
 * launches (terms + partial sums; normalisation once the sums are known).  Limits: G <= 256,
 * K <= 2048. */
int votenet_loss_forward_backward(const VnLossArgs *args, void *stream);

/* size of VnLossArgs.partials in floats: one row of partial sums per workgroup for the masked
 * means of models/loss_helper_labeled.py:70-74, :118-123 and :283-297 */
int votenet_loss_scratch_floats(const VnLossArgs *args);

/* replaces optimizer.step() of torch.optim.Adam(net.parameters(), lr, weight_decay)
 * (pretrain.py:186, :289; train.py:201, :339) on ONE flat parameter buffer, and -- with ema != NULL
 * -- the teacher update that follows it in the semi-supervised stage (train.py:232-236
 * update_ema_variables: teacher = alpha*teacher + (1-alpha)*student, *ema_weight = 1-alpha).
 * p, g, m, v, ema: n floats, 16-byte aligned; step (count of steps so far, incremented), lr and
 * ema_weight: device scalars (a captured graph of the launch replays with current values);
 * g is read as g * grad_scale (the 1/world of a data-parallel mean) and, when grad_scale != 1,
 * rewritten with that product; scratch: 2 floats. */
int votenet_adam_step(long long n, float *p, float *g, float *m, float *v, float *step,
                      const float *lr, double beta1, double beta2, double eps, double weight_decay,
                      double grad_scale, float *ema, const float *ema_weight, float *scratch,
                      void *stream);

/* replaces features.div(torch.norm(features, p=2, dim=1).unsqueeze(1)) on the vote features
 * (models/votenet_iou_branch.py:103-104): x (b,c,n) -> y = x / ||x||_2 over c, norm (b,n) */
int votenet_channel_normalize(int b, int c, int n, const float *x, float *y, float *norm,
                              void *stream);
/* its backward (autograd of models/votenet_iou_branch.py:103-104):
 * dx = (dy - y * sum_c(dy*y)) / norm, from the forward's y and norm */
int votenet_channel_normalize_grad(int b, int c, int n, const float *y, const float *norm,
                                   const float *dy, float *dx, void *stream);

/* replaces decode_scores (models/proposal_module.py:24-54): the proposal head's output net
 * (b, c, k), c = 2 + 3 + 2 nh + 4 ns + nc, split into the named predictions, each a contiguous
 * (b,k,.) tensor: objectness (2), center = agg_xyz + offset (3), heading_scores (nh),
 * heading_residuals_normalized (nh), heading_residuals = normalized * pi/nh (nh), size_scores (ns),
 * size_residuals_normalized = softplus(.) - 1 (ns,3), size_residuals = normalized * mean_size
 * (ns,3), sem_cls_scores (nc) -- one launch instead of five tensor kernels and the strided copies
 * their consumers make. */
int votenet_decode_scores(int b, int k, int nh, int ns, int nc, const float *net,
                          const float *agg_xyz, const float *mean_size, float *objectness,
                          float *center, float *heading_scores, float *heading_resn,
                          float *heading_res, float *size_scores, float *size_resn, float *size_res,
                          float *sem_cls, void *stream);
/* its backward (autograd of models/proposal_module.py:24-54): d_net (b,c,k) from the gradients of
 * the nine outputs (any of them NULL = zero); the gradient of agg_xyz is g_center itself */
int votenet_decode_scores_grad(int b, int k, int nh, int ns, int nc, const float *net,
                               const float *mean_size, const float *g_objectness,
                               const float *g_center, const float *g_heading_scores,
                               const float *g_heading_resn, const float *g_heading_res,
                               const float *g_size_scores, const float *g_size_resn,
                               const float *g_size_res, const float *g_sem_cls, float *d_net,
                               void *stream);

/* decoded boxes of the proposals + one jittered copy of each, training forward (no gradient flows
 * through them): replaces VoteNet.calculate_bbox and the jitter of forward_with_pred_jitter,
 * models/votenet_iou_branch.py:111-137 and :157-172.  center (b,k,3), size_scores (b,k,ns),
 * size_residuals (b,k,ns,3), heading_scores / heading_residuals (b,k,nh), mean_size (ns,3),
 * noise_c / noise_s (b,k,3) standard normal draws -> size (b,k,3) (half sizes), heading (b,k),
 * all_center / all_size (b,2k,3), all_heading (b,2k) = [predicted | jittered], jitter_size2 (b,k,3) */
int votenet_bbox_jitter(int b, int k, int ns, int nh, const float *center, const float *size_scores,
                        const float *size_residuals, const float *heading_scores,
                        const float *heading_residuals, const float *mean_size,
                        const float *noise_c, const float *noise_s, float *size, float *heading,
                        float *all_center, float *all_size, float *all_heading, float *jitter_size2,
                        void *stream);

/* the 4x4x4 grid points of every proposal box and their box-relative coordinates: replaces the
 * linspace / repeat / cat, rot_gpu + torch.bmm and the centre additions of
 * models/grid_conv_module.py:64-83 and the subtraction of :91.  unit (64,3) = the unit grid
 * (x slowest, z fastest); center, size (b,k,3), heading (b,k) -> whole (b,k*64,3), and the
 * relative coordinates into channels 0..2 of feats (b,ctot,k*64) */
int votenet_gridconv_points(int b, int k, int ctot, const float *unit, const float *center,
                            const float *size, const float *heading, float *whole, float *feats,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LOSS_HIP_H */
