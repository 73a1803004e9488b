/*
 * include/mlp_hip.h -- C ABI of the fused BatchNorm(+ReLU)(+max-pool) kernels of the grouped
 * shared MLP (MI355X / gfx950).
 *
 * The reference has no native code for this part of the hot path: it stacks
 * nn.Conv2d(1x1) + nn.BatchNorm2d + nn.ReLU (pointnet2/pytorch_utils.py:14-39,70-124) and pools
 * with F.max_pool2d (pointnet2/pointnet2_modules.py:256-262), i.e. cuDNN/ATen kernels.  These
 * entry points replace the BatchNorm2d / ReLU / max_pool2d launches of one layer (forward and
 * backward); the 1x1 convolution stays a GEMM.
 *
 * Tensors are (B, C, R) float32, contiguous, R = npoint*nsample; per-channel vectors have C
 * entries; pointers are DEVICE pointers; every call returns a hipError_t (0 = success) and runs
 * on the given hipStream_t (void*).  `workspace` holds mlp_bn_workspace_floats(b,c,r) floats.
 */
#ifndef MLP_HIP_H
#define MLP_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scratch (in floats) for the statistics passes below; replaces cuDNN's internal workspace of
 * nn.BatchNorm2d (pytorch_utils.py:42-67) */
size_t mlp_bn_workspace_floats(int b, int c, int r);

/* `tickets` (mlp_bn_train_stats, mlp_bn_relu_backward, mlp_bn_relu_backward_stats,
 * mlp_bn_relu_pool_backward): c ints of device memory OWNED BY THE CALLER, all zero at entry and
 * left all zero at exit.  These launches reduce in one kernel -- every workgroup draws a ticket
 * for its channel and the last one finalizes -- and the library keeps no counters of its own: no
 * global or per-stream state, as the reference's launchers (ball_query_gpu.cu:54: a launch on the
 * caller's stream and nothing else).  Launches ordered one after the other may share an array;
 * launches that may overlap (two streams, two graphs replayed side by side) need one each -- in
 * this package every shared-MLP module owns one (pointnet2/pytorch_utils.py).  NULL is refused.
 * An aborted launch can leave a counter non-zero: zero the array before using it again. */

/* replaces the statistics half of nn.BatchNorm2d in training mode (pytorch_utils.py:53-57 via
 * torch.nn.functional.batch_norm): batch mean / biased variance per channel (Welford-quality:
 * shifted partial sums combined with Chan's formula in double), running_mean / running_var
 * update with `momentum` (unbiased variance, as torch), and the affine coefficients
 * scale = gamma*invstd, shift = beta - mean*scale.  running_* may be NULL. */
int mlp_bn_train_stats(int b, int c, int r, const float *y, const float *gamma, const float *beta,
                       float eps, float momentum, float *running_mean, float *running_var,
                       float *mean, float *invstd, float *scale, float *shift, float *workspace,
                       int *tickets, void *stream);

/* eval-mode coefficients from the running statistics (nn.BatchNorm2d.eval(), pytorch_utils.py:53-57) */
int mlp_bn_eval_coeff(int c, const float *gamma, const float *beta, float eps,
                      const float *running_mean, const float *running_var, float *mean,
                      float *invstd, float *scale, float *shift, void *stream);

/* replaces the normalisation half of nn.BatchNorm2d followed by nn.ReLU (pytorch_utils.py:108-124):
 * z = max(y*scale + shift, 0) */
int mlp_bn_relu_apply(int b, int c, int r, const float *y, const float *scale, const float *shift,
                      float *z, void *stream);

/* replaces BatchNorm2d + ReLU + F.max_pool2d(kernel=[1,nsample]) of the LAST shared-MLP layer
 * (pointnet2_modules.py:256-262): y (b,c,m,ns) -> pooled (b,c,m), plus the arg-max sample and
 * the pre-activation that attained it (kept for the backward pass). */
int mlp_bn_relu_pool(int b, int c, int m, int ns, const float *y, const float *scale,
                     const float *shift, float *pooled, int *argmax, float *ymax, void *stream);

/* replaces the autograd backward of ReLU + BatchNorm2d (pytorch_utils.py:108-124): given dz,
 * returns dy (to be fed to the convolution backward), dgamma, dbeta.  The ReLU mask and the
 * normalised activation are recomputed from y.  training != 0: batch statistics take part in
 * the gradient; training == 0: statistics are constants.  coef: 3*c floats of scratch. */
int mlp_bn_relu_backward(int b, int c, int r, int training, const float *y, const float *dz,
                         const float *gamma, const float *scale, const float *shift,
                         const float *mean, const float *invstd, float *dy, float *dgamma,
                         float *dbeta, float *coef, float *workspace, int *tickets, void *stream);

/* replaces the autograd backward of max_pool2d + ReLU + BatchNorm2d of the last layer
 * (pointnet2_modules.py:256-262): dpooled (b,c,m) -> dy (b,c,m,ns), dgamma, dbeta.  dy == NULL:
 * only dgamma, dbeta and coef are produced (dy is then formed inside mlp_gemm_*_pooled). */
int mlp_bn_relu_pool_backward(int b, int c, int m, int ns, int training, const float *y,
                              const float *dpooled, const int *argmax, const float *ymax,
                              const float *gamma, const float *scale, const float *shift,
                              const float *mean, const float *invstd, float *dy, float *dgamma,
                              float *dbeta, float *coef, float *workspace, int *tickets,
                              void *stream);

/* as mlp_bn_relu_backward but only the per-channel results (dgamma, dbeta, coef = a, c1, c2);
 * the dy tensor itself is then formed inside the operand loads of mlp_gemm_dgrad / _wgrad
 * (replaces the same autograd nodes, pytorch_utils.py:108-124) */
int mlp_bn_relu_backward_stats(int b, int c, int r, int training, const float *y, const float *dz,
                               const float *gamma, const float *scale, const float *shift,
                               const float *mean, const float *invstd, float *dgamma, float *dbeta,
                               float *coef, float *workspace, int *tickets, void *stream);

/* ---- the 1x1 convolution itself, on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32) ----
 * X (b,k,r), W (m,k) row-major, Y (b,m,r).  Replaces nn.Conv2d(kernel 1x1, no bias) of a
 * shared-MLP layer (pytorch_utils.py:70-124, built at :14-39) and its autograd backward. */

/* forward: y = W * x, where x is the given tensor (mode 0) or relu(x*scale[k] + shift[k])
 * (mode 1: the previous layer's BatchNorm+ReLU applied on the fly, pytorch_utils.py:108-124) */
int mlp_gemm_forward(int b, int m, int k, int r, const float *w, const float *x, int mode,
                     const float *scale, const float *shift, float *y, void *stream);

/* BatchNorm statistics as a by-product of the convolution (nn.Conv2d + nn.BatchNorm2d of a
 * shared-MLP layer, pytorch_utils.py:70-124, in training mode): the GEMM epilogue reduces every
 * output channel over the tile's columns to a (mean, M2) pair, so the statistics pass does not
 * re-read y.  mlp_gemm_forward_stats_parts: pairs per channel for this shape (0 = not covered:
 * use mlp_bn_train_stats) and the columns each pair covers. */
int mlp_gemm_forward_stats_parts(int b, int m, int k, int r, int *cols_per_part);
/* mlp_gemm_forward (pytorch_utils.py:70-124) + pairs (parts x m x 2 floats) */
int mlp_gemm_forward_stats(int b, int m, int k, int r, const float *w, const float *x, int mode,
                           const float *scale, const float *shift, float *y, float *pairs,
                           void *stream);
/* mlp_bn_train_stats (nn.BatchNorm2d training statistics, pytorch_utils.py:14-39) from the pairs;
 * scratch: mlp_bn_finalize_pairs_scratch_bytes(c) bytes of device memory */
int mlp_bn_finalize_pairs(int c, int parts, int n_part, const float *pairs, const float *gamma,
                          const float *beta, float eps, float momentum, float *running_mean,
                          float *running_var, float *mean, float *invstd, float *scale,
                          float *shift, void *scratch, void *stream);
/* scratch size of mlp_bn_finalize_pairs (nn.BatchNorm2d keeps no scratch, pytorch_utils.py:14-39) */
size_t mlp_bn_finalize_pairs_scratch_bytes(int c);

/* input gradient: dx (b,k,r) = W^T * dy; wt is W^T (k,m) row-major.  mode 0: dy is given;
 * mode 2: dy is formed on the fly from (y, dz) and the vectors of mlp_bn_relu_backward_stats
 * (replaces conv2d backward-data + the BatchNorm/ReLU backward, pytorch_utils.py:70-124) */
int mlp_gemm_dgrad(int b, int m, int k, int r, const float *wt, int mode, const float *dy,
                   const float *y, const float *dz, const float *scale, const float *shift,
                   const float *mean, const float *invstd, const float *coef, float *dx,
                   void *stream);

/* mlp_gemm_dgrad with the weight as stored (w (m,k) row-major) instead of its transposed copy:
 * the kernel reads A transposed with a swapped lane mapping (replaces conv2d backward-data,
 * pytorch_utils.py:70-124, without the per-call transpose of the weight) */
int mlp_gemm_dgrad_nt(int b, int m, int k, int r, const float *w, int mode, const float *dy,
                      const float *y, const float *dz, const float *scale, const float *shift,
                      const float *mean, const float *invstd, const float *coef, float *dx,
                      void *stream);

/* mlp_gemm_dgrad_pooled with the weight as stored (replaces F.max_pool2d backward + ReLU /
 * BatchNorm2d backward + conv2d backward-data, pointnet2_modules.py:256-262,
 * pytorch_utils.py:70-124) */
int mlp_gemm_dgrad_pooled_nt(int b, int m, int k, int groups, int ns, const float *w, const float *y,
                             const float *dpooled, const int *argmax, const float *scale,
                             const float *shift, const float *mean, const float *invstd,
                             const float *coef, float *dx, void *stream);

/* mlp_gemm_dgrad for the pooled last layer of a set-abstraction MLP: dy (b,m,groups,ns) is formed
 * on the fly from y, dpooled (b,m,groups), argmax (b,m,groups) and the vectors of
 * mlp_bn_relu_pool_backward(dy = NULL) -- neither dz nor dy is written to memory (replaces
 * F.max_pool2d backward + ReLU/BatchNorm2d backward + conv2d backward-data,
 * pointnet2_modules.py:256-262 and pytorch_utils.py:70-124) */
int mlp_gemm_dgrad_pooled(int b, int m, int k, int groups, int ns, const float *wt, const float *y,
                          const float *dpooled, const int *argmax, const float *scale,
                          const float *shift, const float *mean, const float *invstd,
                          const float *coef, float *dx, void *stream);

/* weight gradient: dw (m,k) = sum_b dy[b] * x[b]^T; dy given (pmode 0) or on the fly (pmode 2);
 * x given (qmode 0) or relu(bn(.)) of the previous layer's output (qmode 1).  workspace:
 * mlp_gemm_wgrad_workspace_floats floats (replaces conv2d backward-weight, pytorch_utils.py:70-124) */
int mlp_gemm_wgrad(int b, int m, int k, int r, int pmode, const float *dy, const float *y,
                   const float *dz, const float *scale, const float *shift, const float *mean,
                   const float *invstd, const float *coef, int qmode, const float *x,
                   const float *xscale, const float *xshift, float *dw, float *workspace,
                   void *stream);
/* mlp_gemm_wgrad for the pooled last layer (dy formed on the fly as in mlp_gemm_dgrad_pooled;
 * replaces F.max_pool2d backward + ReLU/BatchNorm2d backward + conv2d backward-weight,
 * pointnet2_modules.py:256-262 and pytorch_utils.py:70-124); workspace as for r = groups*ns */
int mlp_gemm_wgrad_pooled(int b, int m, int k, int groups, int ns, const float *y,
                          const float *dpooled, const int *argmax, const float *scale,
                          const float *shift, const float *mean, const float *invstd,
                          const float *coef, int qmode, const float *x, const float *xscale,
                          const float *xshift, float *dw, float *workspace, void *stream);

/* ---- max over nsample without a second pass over the last layer's output ---------------------
 * relu(y*scale + shift) is monotone in y per channel, so the max-pool over nsample that follows
 * the last layer of an SA module (pointnet2_modules.py:256-262: F.max_pool2d over the nsample
 * axis) is the transform of the group's largest (scale >= 0) or smallest raw value, and
 * sign(scale) = sign(gamma) is known before the batch statistics are: the forward GEMM can leave
 * the winner behind. */
/* 1 when mlp_gemm_forward_stats_pool covers the layer: m 128 or 256, ns 16 / 32 / 64, k % 4 == 0
 * and the epilogue statistics available (dispatch helper for pointnet2_modules.py:256-262) */
int mlp_gemm_forward_stats_pool_supported(int b, int m, int k, int r, int ns);
/* mlp_gemm_forward_stats (mode 1) that also writes ext: 2 planes of (b, m, r/ns) -- the raw output
 * that wins the pool per channel and group of ns columns (the largest where gamma >= 0, the
 * smallest where gamma < 0: gamma = the weight of the BatchNorm that follows) and its first
 * index (replaces the read of y in the max-pool of pointnet2_modules.py:256-262).  y may be NULL:
 * the raw output is then not stored at all (its backward: mlp_pool_gram256_backward) */
int mlp_gemm_forward_stats_pool(int b, int m, int k, int r, const float *w, const float *x,
                                const float *scale, const float *shift, float *y, float *pairs,
                                int ns, const float *gamma, float *ext, void *stream);
/* ---- a set-abstraction module's shared MLP as a register chain (csrc/mlp_chain.hip) -----------
 * conv(1x1) -> BatchNorm -> ReLU -> conv(1x1) -> [statistics, max over nsample] for the SA1 shape
 * 4 -> 64 -> 64 -> 128 (pytorch_utils.py:14-39,70-124; the max-pool of pointnet2_modules.py:256-262):
 * layer 2's activation never leaves the registers between the two GEMMs, BatchNorm statistics and
 * the pooled extrema are in-lane reductions of the second GEMM's accumulators.
 * A workgroup covers 256 columns (4 waves x 2 tiles of 32): many small workgroups, one equal-count
 * (mean, M2) pair per workgroup and channel.
 * mlp_chain_lin4_parts: pairs per channel of a pass and the columns each covers; 0 = shape not
 * covered (m3 = 128, ns 16 / 32 / 64, r % 256 == 0). */
int mlp_chain_lin4_parts(int b, int r, int m3, int ns, int *cols_per_part);
/* bytes of the weight-image scratch of mlp_chain_lin4_prepare (nn.Conv2d weights of
 * pytorch_utils.py:70-124 as fragment-ordered bf16 images) */
size_t mlp_chain_lin4_image_bytes(void);
/* once per forward: w1 (64,4), sc1 / sh1 (64: layer 1's BatchNorm, mlp_first4_bn), w2 (64,64),
 * w3 (128,64) -> img (16-byte aligned, mlp_chain_lin4_image_bytes()); the conv weights of
 * pytorch_utils.py:14-39 */
int mlp_chain_lin4_prepare(const float *w1, const float *sc1, const float *sh1, const float *w2,
                           const float *w3, void *img, void *stream);
/* statistics pass: pairs2 (parts, 64, 2) of y2 = w2 . relu(bn1(w1 . x4)), x4 (b,4,r); nothing else
 * is written (the BatchNorm of pytorch_utils.py:42-67 needs them before layer 2's ReLU) */
int mlp_chain_lin4_stats(int b, int r, int ns, const float *x4, const void *img, float *pairs2,
                         void *stream);
/* full pass: y2 (b,64,r) and y3 (b,128,r) (either may be NULL), pairs3 (parts,128,2), ext = 2 planes
 * of (b,128,r/ns) as mlp_gemm_forward_stats_pool leaves them (pointnet2_modules.py:256-262) */
int mlp_chain_lin4_forward(int b, int r, int ns, const float *x4, const void *img, const float *sc2,
                           const float *sh2, const float *gamma3, float *y2, float *y3, float *pairs3,
                           float *ext, void *stream);
/* training-mode BatchNorm coefficients + running statistics of c channels from a pass's equal-count
 * pairs, n_part columns each (nn.BatchNorm2d inside pytorch_utils.py:42-67) */
int mlp_chain_finalize(int c, int parts, int n_part, const float *pairs, const float *gamma,
                       const float *beta, float eps, float momentum, float *running_mean,
                       float *running_var, float *mean, float *invstd, float *scale, float *shift,
                       void *stream);
/* ---- backward of a max-pooled last layer WITHOUT its raw output (csrc/mlp_pool_gram.hip) -------
 * For (m, k) = (128, 64) (SA1's last layer; pytorch_utils.py:14-39,70-124 with the max-pool of
 * pointnet2_modules.py:256-262): dy3 = q y3 + p + S with y3 = w3 a2 turns both backward products
 * into functions of the layer's input a2 = relu(bn2(y2)), the 64 x 64 matrices W3^T diag(q) W3 and
 * a2 a2^T, and one sparse column per (channel, group) -- y3 is neither read nor stored.
 * mlp_pool_gram_supported: 1 when the layer is covered (ns 16 / 32 / 64, r % 32 == 0). */
int mlp_pool_gram_supported(int b, int m, int k, int r, int ns);
/* per-workgroup partials = parts of stats_part (64, parts, 2) (the autograd of nn.BatchNorm2d,
 * pytorch_utils.py:42-67) */
int mlp_pool_gram_parts(int b, int r);
/* floats of 16-byte aligned workspace of mlp_pool_gram_backward (autograd of nn.Conv2d,
 * pytorch_utils.py:70-124) */
size_t mlp_pool_gram_workspace_floats(int b, int r);
/* dq (b,64,r) = gradient w.r.t. relu(bn2(y2)), dw3 (128,64), stats_part (64,parts,2) = layer 2's
 * BatchNorm-backward sums; y2 (b,64,r) raw, (sc2, sh2, mean2, invstd2) layer 2's BatchNorm,
 * coef3 (128,3) / (sc3, sh3, mean3, invstd3) layer 3's, argmax / dpooled / ymax (b,128,r/ns) the
 * pooled tensors (autograd of pytorch_utils.py:14-39 + pointnet2_modules.py:256-262) */
int mlp_pool_gram_backward(int b, int r, int ns, const float *w3, const float *y2, const float *sc2,
                           const float *sh2, const float *mean2, const float *invstd2,
                           const float *coef3, const float *sc3, const float *sh3, const float *mean3,
                           const float *invstd3, const int *argmax, const float *dpooled,
                           const float *ymax, float *dq, float *dw3, float *stats_part,
                           float *workspace, void *stream);
/* The same for (m, k) = (256, 128) -- the last layer of SA2 / SA3 / SA4 (pointnet2_modules.py:256-262
 * after pytorch_utils.py:14-39,70-124), ns 16 / 32 (csrc/mlp_pool_gram256.hip: two passes over y2,
 * data gradient + BatchNorm-backward sums of the layer below, then weight-gradient sums).
 * mlp_pool_gram256_parts: parts of stats_part (128, parts, 2). */
int mlp_pool_gram256_supported(int b, int m, int k, int r, int ns);
int mlp_pool_gram256_parts(int b, int r);
size_t mlp_pool_gram256_workspace_floats(int b, int r, int ns);
/* dq (b,128,r), dw3 (256,128), stats_part (128,parts,2); y2 (b,128,r) raw; coef3 (256,3); argmax /
 * dpooled / ymax (b,256,r/ns) (autograd of pytorch_utils.py:14-39 + pointnet2_modules.py:256-262) */
int mlp_pool_gram256_backward(int b, int r, int ns, const float *w3, const float *y2, const float *sc2,
                              const float *sh2, const float *mean2, const float *invstd2,
                              const float *coef3, const float *sc3, const float *sh3,
                              const float *mean3, const float *invstd3, const int *argmax,
                              const float *dpooled, const float *ymax, float *dq, float *dw3,
                              float *stats_part, float *workspace, void *stream);
/* pooled, argmax, ymax (b,c,groups) as mlp_bn_relu_pool returns them, from ext
 * (replaces F.max_pool2d of pointnet2_modules.py:256-262 after BatchNorm + ReLU) */
int mlp_bn_pool_from_extrema(int b, int c, int groups, const float *ext, const float *scale,
                             const float *shift, float *pooled, int *argmax, float *ymax,
                             void *stream);

/* ---- dgrad + wgrad of one layer in one pass over its activations -----------------------------
 * The two backward GEMMs of a conv(1x1)+BN+ReLU layer (pytorch_utils.py:70-124: the autograd of
 * nn.Conv2d inside SharedMLP, :14-39) both consume the BatchNorm/ReLU backward of the incoming
 * gradient; run separately (mlp_gemm_dgrad_nt + mlp_gemm_wgrad above) both read the pair it is
 * formed from.  Covered: (m,k) in {(64,64), (128,64), (128,128), (256,128), (128,131), (128,259)},
 * r a multiple of 64 / 32, pmode 2 (from y, dz) or 3 (pooled last layer: (128,64), (256,128),
 * (128,128)),
 * qmode 1 (x = raw output of the previous layer) or 0 (grouped network input: the k = 3+32j
 * shapes). */
/* 1 when mlp_gemm_backward_fused covers the layer (replaces nothing by itself: dispatch helper
 * for the conv backward of pytorch_utils.py:70-124); ns = nsample for pmode 3, else 0 */
int mlp_gemm_backward_fused_supported(int b, int m, int k, int r, int pmode, int qmode, int ns);
/* scratch (floats): one partial dW per persistent workgroup (replaces cuDNN's backward-weight
 * workspace behind pytorch_utils.py:70-124) */
size_t mlp_gemm_backward_fused_workspace_floats(int b, int m, int k, int r);
/* dq (b,k,r) = w^T * P[b] and dw (m,k) = sum_b P[b] * Q[b]^T (replaces conv2d backward-input AND
 * backward-weight, pytorch_utils.py:70-124).  P as in mlp_gemm_dgrad_nt (pmode 2: y, dz (b,m,r))
 * or mlp_gemm_dgrad_pooled_nt (pmode 3: y, dz = dpooled (b,m,r/ns), argmax); Q as in
 * mlp_gemm_wgrad (qmode 1: relu(x*xscale + xshift), qmode 0: x; qmode 4, (64,64) only: the layer
 * below is a 4 -> 64 first layer whose output is not stored -- x is ITS input (b,4,r), xlin_w its
 * weight (64,4), and Q = relu((xlin_w . x)*xscale + xshift) is recomputed.  With qmode 4 the data
 * gradient is NOT written -- its only readers are the virtual layer's BatchNorm sums (stats_part) and
 * weight gradient: dq then receives the gated sums G = sum_n [gate] dq x^T of that layer as parts
 * partials of (64,4) floats, parts = mlp_gemm_backward_fused_stats_parts(): the input of
 * mlp_wgrad_first4_from_gated).
 * dq == NULL (only (128,259)): the weight gradient alone, for a first layer whose input needs no
 * gradient.
 * qmode 1 also needs xmean / xinvstd of the layer that produced x, and, for the k = 64 shapes when
 * stats_part is not NULL, leaves that layer's BatchNorm-backward sums there: stats_part (k, parts, 2) =
 * (sum g, sum g*xhat) with g = dq * [x*xscale + xshift > 0], parts =
 * mlp_gemm_backward_fused_stats_parts() -- the input of mlp_bn_backward_finalize, in place of
 * that layer's mlp_bn_relu_backward_stats pass over (x, dq). */
/* 1: qmode 4 leaves the gated sums in dq (above); 0 (MLP_LIN4_GATED=0): dq (b,64,r) is written as for
 * the other modes and mlp_wgrad_first4 reads it (the round-5 form, pytorch_utils.py:70-124 unchanged) */
int mlp_gemm_backward_fused_lin4_gated(void);
int mlp_gemm_backward_fused(int b, int m, int k, int r, const float *w, int pmode, const float *y,
                            const float *dz, const int *argmax, int ns, const float *scale,
                            const float *shift, const float *mean, const float *invstd,
                            const float *coef, int qmode, const float *x, const float *xscale,
                            const float *xshift, const float *xmean, const float *xinvstd,
                            const float *xlin_w, float *dq, float *dw, float *workspace,
                            float *stats_part, void *stream);
/* The same two GEMMs for the SMALL layers (FP modules, vote / proposal / IoU heads, the pre-gather
 * first layers: b*r <= 16384 columns): conv backward-input and backward-weight of
 * pytorch_utils.py:70-124 are independent and neither fills the chip, so one launch runs both
 * (about the larger of the two instead of their sum).  pmode 0: P = dy (b,m,r) given; pmode 2: P
 * formed on the fly from y, dz and the BatchNorm / ReLU backward constants as in
 * mlp_gemm_dgrad_nt; qmode 0 / 1 as in mlp_gemm_wgrad.  dq == NULL: the weight gradient alone.
 * workspace: mlp_gemm_wgrad_workspace_floats(b, m, k, r) floats. */
int mlp_gemm_backward_small_supported(int b, int m, int k, int r, int pmode, int qmode);
int mlp_gemm_backward_small(int b, int m, int k, int r, const float *w, int pmode,
                            const float *dy_or_y, const float *dz, const float *scale,
                            const float *shift, const float *mean, const float *invstd,
                            const float *coef, int qmode, const float *x, const float *xscale,
                            const float *xshift, float *dq, float *dw, float *workspace,
                            void *stream);
/* bf16 images of the weights of the SMALL layers (FP modules, heads, pre-gather first layers;
 * b*r <= 16384 columns).  Their 64 x 64-tile kernels compute fp32 products as six bf16 MFMAs on an
 * exact three-term split of both operands; every one of a layer's ~128 workgroups split the same
 * weight tile again.  mlp_weight_images_build writes that split ONCE per optimizer step for all
 * weights of a table, in the two orders the forward convolution (pytorch_utils.py:70-124) and its
 * backward-input read; mlp_gemm_forward_img / mlp_gemm_backward_small_img are mlp_gemm_forward /
 * mlp_gemm_backward_small that take the image next to the fp32 weight -- same results bit for bit.
 * An image is mlp_weight_image_elems(m, k) 2-byte elements (three planes, both dimensions padded
 * to multiples of 64), 16-byte aligned.  w / m / k / img / img_t of the build call: HOST arrays. */
size_t mlp_weight_image_elems(int m, int k);
int mlp_weight_images_build(int n, const void *const *w, const int *m, const int *k,
                            void *const *img, void *const *img_t, void *stream);
int mlp_gemm_image_supported(int b, int r);
int mlp_gemm_forward_img(int b, int m, int k, int r, const float *w, const void *img, const float *x,
                         int mode, const float *scale, const float *shift, float *y, void *stream);
int mlp_gemm_backward_small_img(int b, int m, int k, int r, const float *w, const void *img_t,
                                int pmode, const float *dy_or_y, const float *dz, const float *scale,
                                const float *shift, const float *mean, const float *invstd,
                                const float *coef, int qmode, const float *x, const float *xscale,
                                const float *xshift, float *dq, float *dw, float *workspace,
                                void *stream);
/* partials per channel in stats_part; 0 when the layer leaves none (sizing helper for the
 * BatchNorm backward of pytorch_utils.py:42-50) */
int mlp_gemm_backward_fused_stats_parts(int b, int m, int k, int r);
/* dgamma, dbeta and the (c,3) coefficients of dy from (s1, s2) partials (c, parts, 2) over `count`
 * elements per channel (the reduction half of BatchNorm2d's backward, pytorch_utils.py:42-50;
 * same outputs as mlp_bn_relu_backward_stats) */
int mlp_bn_backward_finalize(int c, int parts, double count, int training, const float *partial,
                             const float *gamma, const float *invstd, float *dgamma, float *dbeta,
                             float *coef, void *stream);
/* The weight gradients above (mlp_gemm_wgrad*, mlp_gemm_backward_fused) end with a deterministic
 * reduction of per-workgroup partials in `workspace`.  A weight gradient's only reader is the
 * optimizer (conv2d backward-weight, pytorch_utils.py:70-124, then train.py's optimizer.step()), so
 * a training step may queue those ~30 small reductions of a backward pass and run them as ONE
 * launch: mlp_defer_weight_reductions(1) starts queueing (process-wide), mlp_flush_weight_reductions()
 * launches what is queued on the stream it was issued on, mlp_defer_weight_reductions(0) flushes and
 * returns to immediate launches.  While queued, dw is undefined and the call's workspace must stay
 * allocated. */
int mlp_defer_weight_reductions(int on);
int mlp_flush_weight_reductions(void);

/* ---- first layer of a set-abstraction module applied BEFORE the gather ---------------------------
 * QueryAndGroup (pointnet2_utils.py:335-358) writes grouped = [(xyz[idx] - new_xyz)*s ; feat[idx]]
 * (b, 3+c, m, ns) and the module's first Conv2d (pytorch_utils.py:70-124, called from
 * pointnet2_modules.py:230-262) reads it again.  A 1x1 convolution commutes with the gather:
 *   y1[:, j, t] = (W1 . src)[:, idx[j, t]] - (W1[:, :3] . new_xyz*s)[:, j],  src = [xyz*s ; feat],
 * so the GEMM (mlp_gemm_forward) runs over the n points, not the m*ns gathered columns, on the packed
 * operand src_ext (b, 3+c, n+m) = [xyz*s | new_xyz*s ; feat | 0]; z_ext = W1 . src_ext (b, M, n+m). */
/* 1 when the shape is covered (n, m <= 4096, m*ns <= 32768, ns a power of two >= 4, c % 4 == 0;
 * sizing helper for the grouping of pointnet2_utils.py:335-358) */
int mlp_pregather_supported(int b, int c, int n, int m, int ns);
/* src_ext from xyz (b,n,3), new_xyz (b,m,3), features (b,c,n) (the operand of the reference's
 * grouping, pointnet2_utils.py:348-358, before it is gathered) */
int mlp_pregather_pack(int b, int n, int m, int c, float s, const float *xyz, const float *new_xyz,
                       const float *features, float *src_ext, void *stream);
/* d features (b,c,n) out of d src_ext (b, 3+c, n+m) (the backward of the grouping's feature
 * gather, group_points_gpu.cu:48-80, collapses to this copy) */
int mlp_pregather_unpack_grad(int b, int n, int m, int c, const float *dsrc_ext, float *dfeatures,
                              void *stream);
/* y (b,c,m,ns) = z_ext[.., idx] - z_ext[.., n + j]: the first layer's raw output, as the Conv2d of
 * pytorch_utils.py:70-124 on the grouped tensor would give it; pairs (b, c, 2) or NULL: (mean, M2)
 * of every (cloud, channel) row for mlp_bn_finalize_pairs (parts = b, n_part = m*ns) */
int mlp_pregather_forward(int b, int c, int n, int m, int ns, const float *z_ext, const int *idx,
                          float *y, float *pairs, void *stream);
/* dz_ext (b, c, n+m): the gradient of z_ext.  dy = BatchNorm/ReLU backward of (y, dz) on the fly
 * (operands as mlp_gemm_dgrad_nt's pmode 2), scatter-added over idx through `inverse`
 * (pn2_group_inverse_build) into columns < n, minus its per-group sums into columns n + j
 * (replaces conv2d backward-input + group_points_grad, group_points_gpu.cu:48-80) */
int mlp_pregather_backward(int b, int c, int n, int m, int ns, const float *y, const float *dz,
                           const float *scale, const float *shift, const float *mean,
                           const float *invstd, const float *coef, const unsigned *inverse,
                           float *dz_ext, void *stream);

/* ---- weight gradient of a 4 -> 64 first layer without its output ------------------------------
 * SA1's first layer (mlp [1+3, 64, ...], pointnet2_modules.py:230-262 / pytorch_utils.py:70-124):
 * y = w x is a rank-4 function of x, so only the ReLU-gated part of the BatchNorm backward needs
 * the big gradient tensor; the rest follows from the 4x4 second moments of x. */
/* doubles in a moments buffer (sizing helper for the BatchNorm2d of pytorch_utils.py:42-50) */
int mlp_first4_moments_doubles(void);
/* the 4 sums and 10 products of x (b,4,r) as partial rows of doubles (input of the two entry
 * points below; replaces the statistics pass of BatchNorm2d over the first layer's output,
 * pytorch_utils.py:42-50) */
int mlp_first4_moments(int b, int r, const float *x, double *moments, void *stream);
/* training-mode BatchNorm2d (pytorch_utils.py:42-50) of y = w x, w (64,4), from the moments of x
 * over count = b*r columns: mean, invstd, scale, shift (64) and the running-statistics update, as
 * mlp_bn_finalize_pairs -- without y */
int mlp_first4_bn(const double *moments, double count, const float *w, const float *gamma,
                  const float *beta, float eps, float momentum, float *running_mean,
                  float *running_var, float *mean, float *invstd, float *scale, float *shift,
                  void *stream);
/* mlp_gemm_forward_stats for the SECOND layer (64 -> 64) of such a chain: the operand
 * relu(bn(w1 x4)) is recomputed from x4 (b,4,r); the first layer's output is never stored
 * (replaces conv + BatchNorm + ReLU + conv of pytorch_utils.py:14-39 for SA1's first two layers) */
int mlp_gemm_forward_stats_lin4(int b, int r, const float *w, const float *x4, const float *w1,
                                const float *scale, const float *shift, float *y, float *pairs,
                                void *stream);
/* bytes of the workspace of mlp_wgrad_first4 (sizing helper for the backward-weight of
 * pytorch_utils.py:70-124) */
size_t mlp_wgrad_first4_workspace_bytes(int b, int r);
/* dw (64,4) from x (b,4,r), dz (b,64,r) = gradient w.r.t. relu(bn(y)), and the layer's scale,
 * shift, mean, invstd, coef (64,3) as mlp_bn_relu_backward_stats leaves them; y is not read;
 * moments = those of x (mlp_first4_moments) or NULL to compute them here
 * (replaces conv2d backward-weight behind BatchNorm2d + ReLU, pytorch_utils.py:70-124) */
int mlp_wgrad_first4(int b, int r, const float *w, const float *x, const float *dz,
                     const float *scale, const float *shift, const float *mean, const float *invstd,
                     const float *coef, const double *moments, float *dw, void *workspace,
                     void *stream);
/* The same from gated sums formed elsewhere: mlp_gemm_backward_fused with qmode 4 leaves G = sum_n
 * gate dz x^T as `parts` partials of (64,4) in its dq argument instead of writing dz (the virtual
 * first layer of pytorch_utils.py:14-39: nobody else reads that gradient); moments as above
 * (required), workspace 256 floats */
int mlp_wgrad_first4_from_gated(int parts, const float *gpart, const float *w, const float *mean,
                                const float *invstd, const float *coef, const double *moments,
                                float *dw, float *workspace, void *stream);

/* scratch (floats) for mlp_gemm_wgrad: per-slice partial dW tiles (replaces cuDNN's
 * workspace of conv2d backward-weight, pytorch_utils.py:70-124) */
size_t mlp_gemm_wgrad_workspace_floats(int b, int m, int k, int r);

#ifdef __cplusplus
}
#endif
#endif /* MLP_HIP_H */
