"""Supervised VoteNet-IoU losses (stage-1 pretraining and the labeled half of stage 2).

Host-side mirror of the reference models/loss_helper_labeled.py (compute_vote_loss :28-74,
compute_objectness_loss :77-123, compute_box_and_sem_cls_loss :126-297, get_labeled_loss
:300-370), models/loss_helper_iou.py (compute_iou_labels :52-112) and utils/nn_distance.py
(huber_loss :16-33, nn_distance :35-62): same end_points keys in and out, same loss weights.
Device-agnostic.  The IoU labels come from the gfx950 kernel behind
pcdet.ops.iou3d_nms.iou3d_nms_utils.boxes_iou3d_gpu.
"""
import numpy as np
import torch
import torch.nn.functional as F

from pcdet.ops.iou3d_nms.iou3d_nms_utils import boxes_iou3d_gpu, boxes_iou3d_scene_max_gpu


FAR_THRESHOLD = 0.6
NEAR_THRESHOLD = 0.3
GT_VOTE_FACTOR = 3  # GT votes stored per point
OBJECTNESS_CLS_WEIGHTS = [0.2, 0.8]


def huber_loss(error, delta=1.0, target=None):
    """0.5*x^2 for |x| <= delta, delta*(|x| - 0.5*delta) beyond, x = error (- target)
    (utils/nn_distance.py:16-33; the reference spells it with clamp/pow: 0.5*q^2 + delta*(|x| - q),
    q = min(|x|, delta) -- the same function; one kernel here)."""
    if target is None:
        target = torch.zeros((), dtype=error.dtype, device=error.device).expand_as(error)
    return F.huber_loss(error, target, reduction='none', delta=delta)


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    """Chamfer-style nearest neighbours between (B,N,C) and (B,M,C):
    returns dist1 (B,N), idx1 (B,N), dist2 (B,M), idx2 (B,M)."""
    diff = pc1.unsqueeze(2) - pc2.unsqueeze(1)  # (B,N,M,C)
    if l1smooth:
        dist = torch.sum(huber_loss(diff, delta), dim=-1)
    elif l1:
        dist = torch.sum(torch.abs(diff), dim=-1)
    else:
        dist = torch.sum(diff ** 2, dim=-1)
    dist1, idx1 = torch.min(dist, dim=2)
    dist2, idx2 = torch.min(dist, dim=1)
    return dist1, idx1, dist2, idx2


def _sel(t, inds):
    """t[inds, ...]; `inds is None` means every sample is supervised (stage-1 pretraining, and
    what SupervisedStep passes when supervised_mask is all ones): no gather, no scatter-add in
    the backward.  A slice(0, n) selects the first n samples (stage-2 batches put the labeled
    scenes first, train.py:321-325); label tensors that already have batch n pass through."""
    if inds is None:
        return t
    if isinstance(inds, slice):  # labeled-first batch layout: a view, no gather
        return t if t.shape[0] == inds.stop else t[inds]
    return t[inds, ...]


def _masked_mean(values, mask):
    return torch.sum(values * mask) / (torch.sum(mask) + 1e-6)


def _labels(end_points, inds):
    """GT tensors of the supervised samples; empty GT slots get centre -1000 so that they can
    never be matched (loss_helper_iou.py:56-58)."""
    center = _sel(end_points['center_label'], inds)
    empty = (1 - _sel(end_points['box_label_mask'], inds)).unsqueeze(-1).expand(-1, -1, 3).bool()
    center = torch.where(empty, torch.full_like(center, -1000), center)
    return (center, _sel(end_points['heading_class_label'], inds),
            _sel(end_points['heading_residual_label'], inds),
            _sel(end_points['size_class_label'], inds),
            _sel(end_points['size_residual_label'], inds))


def compute_vote_loss(end_points, supervised_inds):
    """A seed inside an object must vote for (one of) its object centre(s): min L1 distance
    between the predicted votes and the 3 stored GT votes, averaged over object seeds."""
    seed_xyz = _sel(end_points['seed_xyz'], supervised_inds)
    b, num_seed = seed_xyz.shape[:2]
    vote_xyz = _sel(end_points['vote_xyz'], supervised_inds)
    seed_inds = _sel(end_points['seed_inds'], supervised_inds).long()
    mask = torch.gather(end_points['vote_label_mask'], 1, seed_inds)
    gt_votes = torch.gather(end_points['vote_label'], 1,
                            seed_inds.view(b, num_seed, 1).expand(-1, -1, 3 * GT_VOTE_FACTOR))
    # L1 distance of the vote to each of the 3 stored GT votes, the smallest one counts
    # (nn_distance(..., l1=True) on (B*S,1,3) x (B*S,3,3) followed by min over the votes)
    gt_votes = gt_votes.view(b, num_seed, 1, GT_VOTE_FACTOR, 3) + seed_xyz.view(b, num_seed, 1, 1, 3)
    votes = vote_xyz.view(b, num_seed, -1, 1, 3)  # vote_factor predictions per seed
    votes_dist = torch.sum(torch.abs(votes - gt_votes), dim=-1)  # (B, S, vote_factor, 3)
    votes_dist = torch.min(votes_dist.flatten(2), dim=2)[0]
    return _masked_mean(votes_dist, mask.float())


def compute_objectness_loss(end_points, supervised_inds):
    """Proposals within 0.3 m of a GT centre are positives, beyond 0.6 m negatives, the rest
    ignored; weighted cross-entropy.  Also returns the nearest-GT assignment."""
    agg = _sel(end_points['aggregated_vote_xyz'], supervised_inds)
    gt_center = _labels(end_points, supervised_inds)[0]
    dist1, ind1, _, _ = nn_distance(agg, gt_center)
    dist = torch.sqrt(dist1 + 1e-6)
    label = (dist < NEAR_THRESHOLD).long()
    mask = ((dist < NEAR_THRESHOLD) | (dist > FAR_THRESHOLD)).float()
    scores = _sel(end_points['objectness_scores'], supervised_inds)
    weights = _objectness_weights(scores.device)
    ce = F.cross_entropy(scores.transpose(2, 1), label, weight=weights, reduction='none')
    return _masked_mean(ce, mask), label, mask, ind1


_WEIGHTS_ON = {}


def _objectness_weights(device):
    """The class weights, uploaded once per device (a host->device copy per step would also be
    illegal inside a HIP-graph capture)."""
    if device not in _WEIGHTS_ON:
        _WEIGHTS_ON[device] = torch.tensor(OBJECTNESS_CLS_WEIGHTS, device=device)
    return _WEIGHTS_ON[device]


def _block_diagonal_max(iou, b, pred_num):
    """(B*P, B*G) all-pairs IoU -> per-scene (max IoU, argmax GT) of shape (B, P):
    loss_helper_iou.py:106-111."""
    best, assignment = iou.view(b * pred_num, b, -1).max(dim=2)
    scene = torch.arange(b, device=iou.device).unsqueeze(1).expand(-1, pred_num).reshape(-1, 1)
    return (best.gather(dim=1, index=scene).view(b, -1).detach(),
            assignment.gather(dim=1, index=scene).view(b, -1))


def _scene_best_iou(pred_bbox, gt_bbox):
    """(B,P,7),(B,G,7) -> (IoU with the best GT box of the same scene (B,P), its index (B,P)).
    On the GPU one kernel over the same-scene pairs; elsewhere the reference's formulation
    (all pairs through boxes_iou3d_gpu, then the block diagonal)."""
    b, pred_num = pred_bbox.shape[:2]
    if pred_bbox.is_cuda:
        best, assignment = boxes_iou3d_scene_max_gpu(pred_bbox.detach(), gt_bbox.detach())
        return best, assignment
    iou = boxes_iou3d_gpu(pred_bbox.view(-1, 7), gt_bbox.view(-1, 7))
    return _block_diagonal_max(iou, b, pred_num)


def _gt_boxes(end_points, inds, config):
    center, h_cls, h_res, s_cls, s_res = _labels(end_points, inds)
    gt_size = config.class2size_gpu(s_cls, s_res)
    gt_angle = config.class2angle_gpu(h_cls, h_res)
    return torch.cat([center, gt_size, -gt_angle[:, :, None]], dim=2)  # heading sign flips


def compute_iou_labels(end_points, unsupervised_inds, pred_votes, pred_center, pred_sem_cls,
                       pred_objectness, pred_heading_scores, pred_heading_residuals,
                       pred_size_scores, pred_size_residuals, config_dict, reverse=False,
                       with_objectness=True, gt_bbox=None):
    """3-D IoU between every decoded prediction and every GT box of its scene."""
    config = config_dict['dataset_config']
    gt_center = _labels(end_points, unsupervised_inds)[0] if with_objectness else None
    h_cls = torch.argmax(pred_heading_scores, -1)
    h_res = torch.gather(pred_heading_residuals, 2, h_cls.unsqueeze(-1)).squeeze(2)
    s_cls = torch.argmax(pred_size_scores, -1)
    s_res = torch.gather(pred_size_residuals, 2,
                         s_cls.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, 1, 3)).squeeze(2)
    objectness_label = None
    if with_objectness:  # (the labeled loss has these from compute_objectness_loss already)
        dist1, _, _, _ = nn_distance(pred_votes, gt_center)
        objectness_label = (torch.sqrt(dist1 + 1e-6) < NEAR_THRESHOLD).long()
    b = pred_center.shape[0]

    if gt_bbox is None:
        gt_bbox = _gt_boxes(end_points, unsupervised_inds, config)
    pred_size = config.class2size_gpu(s_cls.detach(), s_res)
    pred_size = torch.where(pred_size <= 0, torch.full_like(pred_size, 1e-6), pred_size)
    if config.num_heading_bin == 1:
        pred_angle = torch.zeros(pred_size.shape[:2], device=pred_size.device)
    else:
        pred_angle = config.class2angle_gpu(h_cls.detach(), h_res)
    pred_bbox = torch.cat([pred_center, pred_size, -pred_angle[:, :, None]], dim=2)
    end_points['pred_bbox'] = pred_bbox
    pred_num, gt_num = pred_bbox.shape[1], gt_bbox.shape[1]
    if reverse:
        iou = boxes_iou3d_gpu(gt_bbox.view(-1, 7), pred_bbox.view(-1, 7))
        iou = iou.view(b * gt_num, b, -1)
        scene = torch.arange(b, device=iou.device).unsqueeze(1).expand(-1, gt_num * pred_num)
        scene = scene.reshape(-1, 1, pred_num)
        return iou.gather(dim=1, index=scene).view(b, -1, pred_num).detach()
    iou_labels, object_assignment = _scene_best_iou(pred_bbox, gt_bbox)
    return iou_labels, objectness_label, object_assignment


def _select(values, cls):
    """values (B,K,C[,3]) at class cls (B,K): what `sum(values * one_hot(cls))` evaluates to
    (x*1 + zeros == x exactly), as one gather instead of one_hot / repeat / mul / sum."""
    index = cls.view(*cls.shape, *([1] * (values.dim() - 2)))
    index = index.expand(*cls.shape, 1, *values.shape[3:])
    return torch.gather(values, 2, index).squeeze(2)


def compute_box_and_sem_cls_loss(end_points, supervised_inds, dataset_config, config_dict):
    """Centre / heading / size / class / IoU-estimation losses of the supervised samples
    (loss_helper_labeled.py:126-297).  Same arithmetic per term as the reference; organised so
    that the nine averages over positive proposals share ONE masked reduction (the reference
    spends five small kernels on each) and class-selected residuals are gathers."""
    nh = dataset_config.num_heading_bin
    assign = end_points['object_assignment']
    obj = end_points['objectness_label'].float()
    sup = supervised_inds
    terms = {}  # per-proposal (B,K) quantities averaged over the positive proposals

    def pick(key):
        return torch.gather(_sel(end_points[key], sup), 1, assign)

    # centre: chamfer between predicted centres and GT centres
    terms['center1'], _, dist2, _ = nn_distance(
        _sel(end_points['center'], sup), _sel(end_points['center_label'], sup)[:, :, 0:3])
    center_back = _masked_mean(dist2, _sel(end_points['box_label_mask'], sup))

    # heading: class + residual of the assigned GT
    h_cls_label = pick('heading_class_label')
    terms['heading_cls'] = F.cross_entropy(
        _sel(end_points['heading_scores'], sup).transpose(2, 1), h_cls_label, reduction='none')
    h_res_label = pick('heading_residual_label') / (np.pi / nh)
    h_res_pred = _select(_sel(end_points['heading_residuals_normalized'], sup), h_cls_label)
    terms['heading_reg'] = huber_loss(h_res_pred, delta=1.0, target=h_res_label)

    # size: class + normalised residual
    s_cls_label = pick('size_class_label')
    terms['size_cls'] = F.cross_entropy(
        _sel(end_points['size_scores'], sup).transpose(2, 1), s_cls_label, reduction='none')
    s_res_label = torch.gather(_sel(end_points['size_residual_label'], sup), 1,
                               assign.unsqueeze(-1).expand(-1, -1, 3))
    s_res_pred = _select(_sel(end_points['size_residuals_normalized'], sup), s_cls_label)
    mean_size_label = dataset_config.mean_size(s_res_pred.device)[s_cls_label]
    terms['size_reg'] = torch.mean(
        huber_loss(s_res_pred, delta=1.0, target=s_res_label / mean_size_label), -1)

    # semantic class
    sem_label = pick('sem_cls_label')
    sem_scores = _sel(end_points['sem_cls_scores'], sup)
    terms['sem_cls'] = F.cross_entropy(sem_scores.transpose(2, 1), sem_label, reduction='none')
    terms['cls_acc'] = (sem_label == sem_scores.argmax(dim=-1)).float()

    gt_bbox = _gt_boxes(end_points, sup, dataset_config)  # shared by the two IoU terms
    # IoU labels of the decoded predictions, IoU-estimation loss
    iou_labels, _, iou_assignment = compute_iou_labels(
        end_points, sup, _sel(end_points['aggregated_vote_xyz'], sup),
        _sel(end_points['center'], sup), None, None, _sel(end_points['heading_scores'], sup),
        _sel(end_points['heading_residuals'], sup), _sel(end_points['size_scores'], sup),
        _sel(end_points['size_residuals'], sup), config_dict={'dataset_config': dataset_config},
        with_objectness=False, gt_bbox=gt_bbox)
    end_points['pred_iou_value'] = iou_labels.mean()
    terms['pred_iou_obj'] = iou_labels
    if 'iou_scores' in end_points:
        iou_pred = torch.sigmoid(_sel(end_points['iou_scores'], sup))
        if iou_pred.shape[2] > 1:
            iou_sem = torch.gather(_sel(end_points['sem_cls_label'], sup), 1, iou_assignment)
            iou_pred = torch.gather(iou_pred, 2, iou_sem.unsqueeze(-1)).squeeze(-1)
        else:
            iou_pred = iou_pred.squeeze(-1)
        iou_acc = torch.abs(iou_pred - iou_labels)
        end_points['iou_acc'] = iou_acc.mean()
        terms['iou_acc_obj'] = iou_acc
        end_points['iou_loss'] = huber_loss(iou_pred, delta=1.0, target=iou_labels).mean()

    if 'jitter_center' in end_points:
        pred_bbox = torch.cat([_sel(end_points['jitter_center'], sup),
                               _sel(end_points['jitter_size'], sup),
                               -_sel(end_points['jitter_heading'], sup)[:, :, None]], dim=2)
        jitter_iou_labels, jitter_assign = _scene_best_iou(pred_bbox, gt_bbox)
        jitter_sem = torch.gather(_sel(end_points['sem_cls_label'], sup), 1, jitter_assign)
        jitter_pred = torch.sigmoid(_sel(end_points['iou_scores_jitter'], sup))
        jitter_pred = torch.gather(jitter_pred, 2, jitter_sem.unsqueeze(-1)).squeeze(-1) \
            if jitter_pred.shape[2] > 1 else jitter_pred.squeeze(-1)
        jitter_acc = torch.abs(jitter_pred - jitter_iou_labels)
        end_points['jitter_iou_acc'] = jitter_acc.mean()
        end_points['jitter_iou_acc_obj'] = jitter_acc.sum() / (jitter_acc.numel() + 1e-6)
        end_points['jitter_iou_loss'] = huber_loss(
            jitter_pred, delta=1.0, target=jitter_iou_labels).sum() / (jitter_acc.numel() + 1e-6)

    # sum(term * obj) / (sum(obj) + 1e-6) for every term, in one reduction
    names = list(terms)
    count = torch.sum(obj)
    means = torch.sum(torch.stack([terms[n] for n in names]) * obj, dim=(1, 2)) / (count + 1e-6)
    mean = dict(zip(names, means.unbind(0)))
    end_points['obj_count'] = count
    end_points['cls_acc'] = mean['cls_acc']
    end_points['pred_iou_obj_value'] = mean['pred_iou_obj']
    if 'iou_acc_obj' in mean:
        end_points['iou_acc_obj'] = mean['iou_acc_obj']
    return (mean['center1'] + center_back, mean['heading_cls'], mean['heading_reg'],
            mean['size_cls'], mean['size_reg'], mean['sem_cls'])


def get_labeled_loss(end_points, dataset_config, config_dict=None):
    """10 * (vote + 0.5*objectness + box + 0.1*sem_cls + iou [+ jitter_iou]) over the samples
    with supervised_mask == 1; fills end_points with every intermediate loss / statistic."""
    if end_points.get('all_supervised', False):  # host-side knowledge: skip every gather
        supervised_inds = None
    elif end_points.get('labeled_num') is not None:  # labeled scenes first, count known on the host
        supervised_inds = slice(0, int(end_points['labeled_num']))
    else:
        supervised_inds = end_points.get('supervised_inds')  # static under HIP-graph capture
        if supervised_inds is None:
            supervised_inds = torch.nonzero(end_points['supervised_mask']).squeeze(1).long()

    from . import fused_loss
    if fused_loss.enabled() and fused_loss.supported(end_points, supervised_inds) and \
            fused_loss.available(end_points['center'].device):
        return fused_loss.get_labeled_loss_fused(end_points, dataset_config, supervised_inds)

    end_points['vote_loss'] = compute_vote_loss(end_points, supervised_inds)
    objectness_loss, objectness_label, objectness_mask, object_assignment = \
        compute_objectness_loss(end_points, supervised_inds)
    end_points['objectness_loss'] = objectness_loss
    end_points['objectness_label'] = objectness_label
    end_points['objectness_mask'] = objectness_mask
    end_points['object_assignment'] = object_assignment
    total = float(objectness_label.shape[0] * objectness_label.shape[1])
    end_points['pos_ratio'] = torch.sum(objectness_label.float()) / total
    end_points['neg_ratio'] = torch.sum(objectness_mask.float()) / total - end_points['pos_ratio']

    (center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss, size_reg_loss,
     sem_cls_loss) = compute_box_and_sem_cls_loss(end_points, supervised_inds, dataset_config,
                                                  config_dict)
    end_points['center_loss'] = center_loss
    end_points['heading_cls_loss'] = heading_cls_loss
    end_points['heading_reg_loss'] = heading_reg_loss
    end_points['size_cls_loss'] = size_cls_loss
    end_points['size_reg_loss'] = size_reg_loss
    end_points['sem_cls_loss'] = sem_cls_loss
    box_loss = 0.1 * heading_cls_loss + heading_reg_loss + 0.1 * size_cls_loss + size_reg_loss \
        + center_loss
    end_points['box_loss'] = box_loss

    loss = end_points['vote_loss'] + 0.5 * objectness_loss + box_loss + 0.1 * sem_cls_loss
    loss = loss + end_points['iou_loss']
    if 'jitter_iou_loss' in end_points:
        loss = loss + end_points['jitter_iou_loss']
    loss = loss * 10
    end_points['detection_loss'] = loss
    end_points['loss'] = loss

    obj_pred = torch.argmax(_sel(end_points['objectness_scores'], supervised_inds), 2)
    end_points['obj_acc'] = _masked_mean((obj_pred == objectness_label.long()).float(),
                                         objectness_mask)
    return loss, end_points
