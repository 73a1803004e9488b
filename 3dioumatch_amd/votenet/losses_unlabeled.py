"""Consistency loss on the unlabeled half of a semi-supervised batch (stage 2 of 3DIoUMatch).

Host-side mirror of the reference models/loss_helper_unlabeled.py: trans_center :24-36,
trans_size :39-51, trans_angle :54-64, compute_objectness_loss :137-196,
compute_box_and_sem_cls_loss :199-289, get_pseudo_detection_loss :292-361, get_pseudo_labels
:364-538, get_unlabeled_loss :541-600 (the `view_stats` branches, which peek at ground truth for
logging only, are not mirrored) -- same end_points keys, thresholds and loss weights.

What differs is WHERE it runs: the reference's pseudo-label filter leaves the device (per-scene
numpy loops over 64 boxes for get_3d_box + lhs_3d_faster_samecls, boolean-mask assignments that
synchronise); here everything stays on the device with static shapes -- the NMS is one kernel
(votenet/pseudo_nms.py) and every data-dependent selection is a `torch.where` -- so the whole
semi-supervised step can be captured in a HIP graph.  The batch layout is the reference loader's:
labeled samples first, then the unlabeled ones (train.py:321-325), `labeled_num` known on the host.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from .losses import (FAR_THRESHOLD, NEAR_THRESHOLD, _masked_mean, _objectness_weights, huber_loss,
                     nn_distance)

MAX_NUM_OBJ = 64


def _lhs_nms(center, size, heading, score, cls, thresh, old_type):
    """(S,n) bool keep mask of lhs_3d_faster_samecls; the GPU kernel (tests substitute the oracle)."""
    from .pseudo_nms import lhs_nms_samecls_gpu
    return lhs_nms_samecls_gpu(center, size, heading, score, cls, thresh, old_type)


def _flag(t):
    return t.view(-1, 1).bool()


def trans_center(center, flip_x_axis, flip_y_axis, rot_mat, scale_ratio):
    """teacher-frame centres (B,K,3) -> student frame: flips, rotation, scale."""
    x = torch.where(_flag(flip_x_axis), -center[:, :, 0], center[:, :, 0])
    y = torch.where(_flag(flip_y_axis), -center[:, :, 1], center[:, :, 1])
    out = torch.stack([x, y, center[:, :, 2]], dim=2)
    out = torch.bmm(out, rot_mat.transpose(1, 2))
    return out * scale_ratio


def trans_size(size_class, size_residual, scale_ratio, config):
    base = config.mean_size(size_residual.device)[size_class.reshape(-1)].view(*size_class.shape, 3)
    return (base + size_residual) * scale_ratio - base


def trans_angle(angle_class, angle_residual, flip_x_axis, flip_y_axis, rot_angle, config):
    angle = config.class2angle_gpu(angle_class, angle_residual)
    angle = torch.where(_flag(flip_x_axis), np.pi - angle, angle)
    angle = torch.where(_flag(flip_y_axis), -angle, angle)
    angle = angle - rot_angle.unsqueeze(-1)
    new_class, new_residual = config.angle2class_gpu(angle)
    return new_class.long(), new_residual


def get_pseudo_labels(end_points, ema_end_points, labeled_num, pred_center, pred_sem_cls,
                      pred_objectness, pred_heading_scores, pred_heading_residuals,
                      pred_size_scores, pred_size_residuals, pred_vote_xyz, config_dict):
    """Teacher predictions of the unlabeled scenes -> at most 64 pseudo boxes per scene."""
    config = config_dict['dataset_config']
    pred_objectness = torch.softmax(pred_objectness, dim=2)
    pos_obj, neg_obj = pred_objectness[:, :, 1], pred_objectness[:, :, 0]
    objectness_mask = pos_obj > config_dict['obj_threshold']
    neg_objectness_mask = neg_obj > 0.9
    pred_sem_cls = torch.softmax(pred_sem_cls, dim=2)
    max_cls, argmax_cls = torch.max(pred_sem_cls, dim=2)
    cls_mask = max_cls > config_dict['cls_threshold']

    iou_pred = torch.sigmoid(ema_end_points['iou_scores'][labeled_num:])
    if iou_pred.shape[2] > 1:
        iou_pred = torch.gather(iou_pred, 2, argmax_cls.unsqueeze(-1)).squeeze(-1)
    else:
        iou_pred = iou_pred.squeeze(-1)
    final_mask = cls_mask & objectness_mask & (iou_pred > config_dict['iou_threshold'])

    # keep the MAX_NUM_OBJ predictions with the largest pos_obj * max_cls among the survivors
    inds = torch.argsort(pos_obj * max_cls * final_mask, dim=1, descending=True, stable=True)
    inds = inds[:, :MAX_NUM_OBJ]
    inds3 = inds.unsqueeze(-1).expand(-1, -1, 3)
    final_mask_sorted = torch.gather(final_mask, 1, inds)
    end_points['pseudo_gt_ratio'] = final_mask_sorted.float().mean()
    neg_objectness_mask = torch.gather(neg_objectness_mask, 1, inds)

    argmax_size = torch.argmax(pred_size_scores, dim=2)
    argmax_heading = torch.argmax(pred_heading_scores, dim=2)
    pred_heading_residuals = torch.gather(pred_heading_residuals, 2,
                                          argmax_heading.unsqueeze(-1)).squeeze(2)
    pred_size_residuals = torch.gather(
        pred_size_residuals, 2, argmax_size.view(*argmax_size.shape, 1, 1).expand(-1, -1, -1, 3)
    ).squeeze(2)

    heading_label = torch.gather(argmax_heading, 1, inds)
    heading_residual_label = torch.gather(pred_heading_residuals, 1, inds)
    size_label = torch.gather(argmax_size, 1, inds)
    size_residual_label = torch.gather(pred_size_residuals, 1, inds3)
    sem_cls_label = torch.gather(argmax_cls, 1, inds)
    center_label = torch.gather(pred_center, 1, inds3)
    iou_label = torch.gather(iou_pred, 1, inds)

    if config_dict['use_lhs']:
        # numpy decoding of the reference: float64 mean sizes / class angles + float32 residuals
        size64 = config.mean_size(center_label.device).double()[size_label] \
            + size_residual_label.double()
        heading64 = config.class2angle_f64(heading_label, heading_residual_label)
        score = torch.gather(pos_obj, 1, inds) * iou_label
        keep = _lhs_nms(center_label.detach().contiguous(), size64.detach(), heading64.detach(),
                        score.detach().contiguous(), sem_cls_label, config_dict['nms_iou'],
                        config_dict['use_old_type_nms'])
        final_mask_sorted = final_mask_sorted & keep

    label_mask = final_mask_sorted.long()
    gone = torch.full_like(center_label, -1000)
    center_label = torch.where(final_mask_sorted.unsqueeze(-1), center_label, gone)
    false_center_label = torch.where(neg_objectness_mask.unsqueeze(-1),
                                     torch.gather(pred_vote_xyz, 1, inds3), gone)
    return (label_mask, center_label, sem_cls_label, heading_label, heading_residual_label,
            size_label, size_residual_label, false_center_label, iou_label)


def compute_objectness_loss(end_points, labeled_num):
    """Near/far assignment of the student's proposals to the pseudo boxes
    (loss_helper_unlabeled.py:137-196, samecls_match=False)."""
    agg = end_points['aggregated_vote_xyz'][labeled_num:]
    gt_center = end_points['unlabeled_center_label'][:, :, 0:3]
    empty = (1 - end_points['unlabeled_box_label_mask']).unsqueeze(-1).bool()
    gt_center = torch.where(empty, torch.full_like(gt_center, -1000), gt_center)
    end_points['unlabeled_center_label'] = gt_center  # the reference masks it in place
    dist1, ind1, _, _ = nn_distance(agg, gt_center)
    dist = torch.sqrt(dist1 + 1e-6)
    label = (dist < NEAR_THRESHOLD).long()
    mask = ((dist < NEAR_THRESHOLD) | (dist > FAR_THRESHOLD)).float()
    scores = end_points['objectness_scores'][labeled_num:]
    ce = F.cross_entropy(scores.transpose(2, 1), label, weight=_objectness_weights(scores.device),
                         reduction='none')
    return _masked_mean(ce, mask), label, mask, ind1


def compute_box_and_sem_cls_loss(end_points, labeled_num, config):
    """Box + class losses of the student against the pseudo boxes (:199-289)."""
    nh, ns = config.num_heading_bin, config.num_size_cluster
    assign = end_points['unlabeled_object_assignment']
    box_label_mask = end_points['unlabeled_box_label_mask'].float()
    obj = end_points['unlabeled_objectness_label'].float()

    def pick(key):
        return torch.gather(end_points[key], 1, assign)

    dist1, _, dist2, _ = nn_distance(end_points['center'][labeled_num:],
                                     end_points['unlabeled_center_label'][:, :, 0:3])
    center_loss = _masked_mean(dist1, obj) + _masked_mean(dist2, box_label_mask)

    h_cls_label = pick('unlabeled_heading_class_label')
    heading_class_loss = _masked_mean(
        F.cross_entropy(end_points['heading_scores'][labeled_num:].transpose(2, 1), h_cls_label,
                        reduction='none'), obj)
    h_res_label = pick('unlabeled_heading_residual_label') / (np.pi / nh)
    h_onehot = F.one_hot(h_cls_label, nh).float()
    h_res_pred = torch.sum(end_points['heading_residuals_normalized'][labeled_num:] * h_onehot, -1)
    heading_reg_loss = _masked_mean(huber_loss(h_res_pred, delta=1.0, target=h_res_label), obj)

    s_cls_label = pick('unlabeled_size_class_label')
    size_class_loss = _masked_mean(
        F.cross_entropy(end_points['size_scores'][labeled_num:].transpose(2, 1), s_cls_label,
                        reduction='none'), obj)
    s_res_label = torch.gather(end_points['unlabeled_size_residual_label'], 1,
                               assign.unsqueeze(-1).repeat(1, 1, 3))
    s_onehot = F.one_hot(s_cls_label, ns).float().unsqueeze(-1).repeat(1, 1, 1, 3)
    s_res_pred = torch.sum(end_points['size_residuals_normalized'][labeled_num:] * s_onehot, 2)
    mean_size = config.mean_size(s_res_pred.device).unsqueeze(0).unsqueeze(0)
    mean_size_label = torch.sum(s_onehot * mean_size, 2)
    size_reg_loss = _masked_mean(
        torch.mean(huber_loss(s_res_pred, delta=1.0, target=s_res_label / mean_size_label), -1), obj)

    sem_label = pick('unlabeled_sem_cls_label')
    sem_cls_loss = _masked_mean(
        F.cross_entropy(end_points['sem_cls_scores'][labeled_num:].transpose(2, 1), sem_label,
                        reduction='none'), obj)
    return (center_loss, heading_class_loss, heading_reg_loss, size_class_loss, size_reg_loss,
            sem_cls_loss)


def get_pseudo_detection_loss(end_points, labeled_num, config):
    """10 * (box + 0.1 * sem_cls) on the unlabeled samples (:292-361).  On the GPU: the kernels of
    the supervised loss in their consistency mode (fused_loss.get_pseudo_detection_loss_fused;
    VOTENET_FUSED_LOSS=0 or VOTENET_FUSED_CONSISTENCY=0 keep the tensor operations below)."""
    from . import fused_loss
    if fused_loss.enabled() and os.environ.get("VOTENET_FUSED_CONSISTENCY", "1") != "0" and \
            fused_loss.available(end_points['center'].device) and end_points['center'].dim() == 3:
        return fused_loss.get_pseudo_detection_loss_fused(end_points, labeled_num, config)
    obj_loss, obj_label, obj_mask, assignment = compute_objectness_loss(end_points, labeled_num)
    end_points['unlabeled_objectness_loss'] = obj_loss
    end_points['unlabeled_objectness_label'] = obj_label
    end_points['unlabeled_objectness_mask'] = obj_mask
    end_points['unlabeled_object_assignment'] = assignment
    total = float(obj_label.shape[0] * obj_label.shape[1])
    end_points['unlabeled_pos_ratio'] = torch.sum(obj_label.float()) / total
    end_points['unlabeled_neg_ratio'] = torch.sum(obj_mask) / total - end_points['unlabeled_pos_ratio']

    (center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss, size_reg_loss,
     sem_cls_loss) = compute_box_and_sem_cls_loss(end_points, labeled_num, config)
    end_points['unlabeled_center_loss'] = center_loss
    end_points['unlabeled_heading_cls_loss'] = heading_cls_loss
    end_points['unlabeled_heading_reg_loss'] = heading_reg_loss
    end_points['unlabeled_size_cls_loss'] = size_cls_loss
    end_points['unlabeled_size_reg_loss'] = size_reg_loss
    end_points['unlabeled_sem_cls_loss'] = sem_cls_loss
    box_loss = 0.1 * heading_cls_loss + heading_reg_loss + 0.1 * size_cls_loss + size_reg_loss \
        + center_loss
    end_points['unlabeled_box_loss'] = box_loss
    loss = (box_loss + 0.1 * sem_cls_loss) * 10
    end_points['unlabeled_detection_loss'] = loss
    return loss, end_points


def default_config_dict(config, dataset='scannet', unlabeled_batch_size=8):
    """The filter settings of train.py:263-275."""
    return {'dataset_config': config, 'unlabeled_batch_size': unlabeled_batch_size,
            'dataset': dataset, 'nms_iou': 0.25, 'use_old_type_nms': False, 'obj_threshold': 0.9,
            'cls_threshold': 0.9, 'use_lhs': True, 'iou_threshold': 0.25, 'samecls_match': False,
            'view_stats': False}


def get_unlabeled_loss(end_points, ema_end_points, config, config_dict, labels_only=False):
    """Pseudo labels from the EMA teacher -> transformed into the student's augmented frame ->
    consistency loss.  `labeled_num`: end_points['labeled_num'] (host int) or, as the reference
    does, the number of non-zero entries of supervised_mask (a device sync)."""
    labeled_num = end_points.get('labeled_num')
    if labeled_num is None:
        labeled_num = int(torch.count_nonzero(end_points['supervised_mask']))
    tail = slice(labeled_num, None)
    aug = [end_points[k][tail] for k in ('flip_x_axis', 'flip_y_axis', 'rot_mat', 'scale')]
    fused = None
    if ema_end_points['center'].is_cuda and os.environ.get("VOTENET_FUSED_PSEUDO_LABELS", "1") != "0":
        from . import pseudo_nms
        teacher = [ema_end_points[k][tail] for k in (
            'objectness_scores', 'sem_cls_scores', 'iou_scores', 'heading_scores', 'heading_residuals',
            'size_scores', 'size_residuals', 'aggregated_vote_xyz')]
        if pseudo_nms.pseudo_labels_supported(ema_end_points['center'][tail], aug[3], f32=teacher + [aug[2]],
                                              i64=aug[:2], sem_cls=teacher[1], iou_scores=teacher[2]) and \
                _lhs_nms.__module__ == __name__:  # (tests substitute the NMS: the tensor form calls it)
            fused = pseudo_nms.pseudo_labels_gpu(
                ema_end_points['objectness_scores'][tail], ema_end_points['sem_cls_scores'][tail],
                ema_end_points['iou_scores'][tail], ema_end_points['heading_scores'][tail],
                ema_end_points['heading_residuals'][tail], ema_end_points['size_scores'][tail],
                ema_end_points['size_residuals'][tail], ema_end_points['center'][tail],
                ema_end_points['aggregated_vote_xyz'][tail], config.mean_size(aug[3].device), aug[0], aug[1],
                aug[2], aug[3], config_dict['obj_threshold'], config_dict['cls_threshold'],
                config_dict['iou_threshold'],
                (config_dict['nms_iou'], config_dict['use_old_type_nms']) if config_dict['use_lhs'] else None)
    if fused is not None:
        # two launches around the NMS kernel: selection, labels, and the transforms into the student's
        # frame (votenet/pseudo_nms.py pseudo_labels_gpu)
        end_points['pseudo_gt_ratio'] = fused['pseudo_gt_ratio']
        label_mask, center_label, false_center_label = (fused[k] for k in ('label_mask', 'center_label',
                                                                            'false_center_label'))
        sem_cls_label, heading_label, size_label = (fused[k] for k in ('sem_cls_label', 'heading_label',
                                                                       'size_label'))
        heading_residual_label, size_residual_label, iou_label = (
            fused[k] for k in ('heading_residual_label', 'size_residual_label', 'iou_label'))
    else:
        (label_mask, center_label, sem_cls_label, heading_label, heading_residual_label, size_label,
         size_residual_label, false_center_label, iou_label) = get_pseudo_labels(
            end_points, ema_end_points, labeled_num, ema_end_points['center'][tail],
            ema_end_points['sem_cls_scores'][tail], ema_end_points['objectness_scores'][tail],
            ema_end_points['heading_scores'][tail], ema_end_points['heading_residuals'][tail],
            ema_end_points['size_scores'][tail], ema_end_points['size_residuals'][tail],
            ema_end_points['aggregated_vote_xyz'][tail], config_dict)
        center_label = trans_center(center_label, *aug)
        false_center_label = trans_center(false_center_label, *aug)
        size_residual_label = trans_size(size_label, size_residual_label, aug[3], config)
    if config_dict['dataset'] == 'sunrgbd':
        heading_label, heading_residual_label = trans_angle(
            heading_label, heading_residual_label, aug[0], aug[1], end_points['rot_angle'][tail],
            config)

    end_points['unlabeled_center_label'] = center_label
    end_points['unlabeled_box_label_mask'] = label_mask
    end_points['unlabeled_sem_cls_label'] = sem_cls_label
    end_points['unlabeled_heading_class_label'] = heading_label
    end_points['unlabeled_heading_residual_label'] = heading_residual_label
    end_points['unlabeled_size_class_label'] = size_label
    end_points['unlabeled_size_residual_label'] = size_residual_label
    end_points['unlabeled_false_center_label'] = false_center_label
    end_points['unlabeled_iou_label'] = iou_label
    if labels_only:  # (the caller computes the loss together with the supervised one)
        return None, end_points
    return get_pseudo_detection_loss(end_points, labeled_num, config)
