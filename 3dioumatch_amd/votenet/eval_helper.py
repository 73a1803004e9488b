"""Prediction parsing of the evaluation path with the NMS on the device.

Host-side mirror of the reference models/ap_helper.py: predictions2corners3d :60-99 and
parse_predictions :101-222 for the 3-D NMS branches the training scripts use
(`use_3d_nms`, with or without `cls_nms`, optionally IoU-weighted scores; train.py:263-275).
The reference decodes every box on the host (B x K calls of get_3d_box,
utils/box_util.py:335-358) and runs utils/nms.py per scene in numpy; here the decoding is a few
batched float64 tensor ops, the NMS one kernel launch (votenet/pseudo_nms.py:nms3d_aabb_gpu) and
only the final, small Python lists are built on the host.

parse_groundtruths (:224-307) and APCalculator (:382-435) complete the evaluation loop of
train.py:evaluate_one_epoch; the average precision itself is votenet/eval_det.py (oriented-box IoU
on the device).

Not mirrored (raise NotImplementedError): `remove_empty_box` (off in the reference's configs) and
the 2-D NMS branch.
"""
import numpy as np
import torch


def _nms3d(center, size, heading, score, cls, thresh, old_type, same_class):
    """(S,n) bool keep mask; the GPU kernel (tests substitute the oracle)."""
    from .pseudo_nms import nms3d_aabb_gpu
    return nms3d_aabb_gpu(center, size, heading, score, cls, thresh, old_type, same_class)


def corners_upright_camera(center, size64, heading64):
    """(B,K,3) f32 centres (depth frame), (B,K,3) f64 sizes (l,w,h), (B,K) f64 heading ->
    (B,K,8,3) f32 corners in the upright camera frame, vertex order and float64 arithmetic of
    get_3d_box (utils/box_util.py:335-358) on flip_axis_to_camera centres (ap_helper.py:28-35)."""
    dev = center.device
    sx = torch.tensor([1, 1, -1, -1, 1, 1, -1, -1], dtype=torch.float64, device=dev)
    sy = torch.tensor([1, 1, 1, 1, -1, -1, -1, -1], dtype=torch.float64, device=dev)
    sz = torch.tensor([1, -1, -1, 1, 1, -1, -1, 1], dtype=torch.float64, device=dev)
    l, w, h = size64[..., 0:1], size64[..., 1:2], size64[..., 2:3]
    x, y, z = sx * l / 2, sy * h / 2, sz * w / 2                      # (B,K,8)
    c, s = torch.cos(heading64).unsqueeze(-1), torch.sin(heading64).unsqueeze(-1)
    cx = center[..., 0:1].double()
    cy = -center[..., 2:3].double()
    cz = center[..., 1:2].double()
    px = (c * x + 0.0 * y + s * z) + cx
    py = (0.0 * x + 1.0 * y + 0.0 * z) + cy
    pz = (-s * x + 0.0 * y + c * z) + cz
    return torch.stack([px, py, pz], dim=-1).float()


def decode_boxes(end_points, config):
    """arg-max heading / size class + residual -> (size64 (B,K,3), heading64 (B,K)), decoded in
    float64 as the reference's numpy class2size / class2angle do (ap_helper.py:77-92)."""
    h_cls = torch.argmax(end_points['heading_scores'], -1)
    h_res = torch.gather(end_points['heading_residuals'], 2, h_cls.unsqueeze(-1)).squeeze(2)
    s_cls = torch.argmax(end_points['size_scores'], -1)
    s_res = torch.gather(end_points['size_residuals'], 2,
                         s_cls.view(*s_cls.shape, 1, 1).expand(-1, -1, -1, 3)).squeeze(2)
    size64 = config.mean_size(s_res.device).double()[s_cls] + s_res.double()
    heading64 = config.class2angle_f64(h_cls, h_res)
    return size64, heading64


@torch.no_grad()
def parse_predictions(end_points, config_dict):
    """-> batch_pred_map_cls: per scene a list of (class, corners (8,3) ndarray, confidence), as
    ap_helper.parse_predictions returns it; also fills end_points['pred_mask'] (B,K) and
    end_points['batch_pred_map_cls']."""
    if config_dict.get('remove_empty_box', False):
        raise NotImplementedError("remove_empty_box is not mirrored (off in the reference configs)")
    if not config_dict.get('use_3d_nms', True):
        raise NotImplementedError("only the 3-D NMS branches are mirrored")
    config = config_dict['dataset_config']
    center = end_points['center']
    sem_probs = torch.softmax(end_points['sem_cls_scores'], dim=-1)
    pred_sem_cls = torch.argmax(end_points['sem_cls_scores'], -1)
    obj_prob = torch.softmax(end_points['objectness_scores'], dim=-1)[:, :, 1]
    size64, heading64 = decode_boxes(end_points, config)
    corners = corners_upright_camera(center, size64, heading64)

    scores = obj_prob
    same_class = bool(config_dict.get('cls_nms', False))
    if same_class and config_dict.get('use_iou_for_nms', False):
        iou = torch.sigmoid(end_points['iou_scores'])
        if iou.shape[2] > 1:
            iou = torch.gather(iou, 2, pred_sem_cls.unsqueeze(-1))
        scores = scores * iou.squeeze(-1)
    pred_mask = _nms3d(center.contiguous(), size64, heading64, scores.contiguous(), pred_sem_cls,
                       config_dict['nms_iou'], config_dict['use_old_type_nms'], same_class)
    # one device->host copy of the small results, then the reference's list layout
    keep = (pred_mask & (obj_prob > config_dict['conf_thresh'])).cpu().numpy()
    # (B,K) float64 numpy array of 0/1 like the reference's (ap_helper.py:141-155), so that
    # consumers such as dump_helper index / multiply it the same way
    end_points['pred_mask'] = pred_mask.cpu().numpy().astype(np.float64)
    corners_h = corners.cpu().numpy()
    obj_h = obj_prob.cpu().numpy()
    sem_h = sem_probs.cpu().numpy()
    cls_h = pred_sem_cls.cpu().numpy()
    batch = []
    for i in range(keep.shape[0]):
        js = np.nonzero(keep[i])[0]
        if config_dict['per_class_proposal']:
            cur = [(ii, corners_h[i, j], sem_h[i, j, ii] * obj_h[i, j])
                   for ii in range(config.num_class) for j in js]
        else:
            cur = [(int(cls_h[i, j]), corners_h[i, j], obj_h[i, j]) for j in js]
        batch.append(cur)
    end_points['batch_pred_map_cls'] = batch
    return batch


@torch.no_grad()
def parse_groundtruths(end_points, config_dict):
    """-> batch_gt_map_cls: per scene a list of (class, corners (8,3) float32 ndarray) of the boxes
    with box_label_mask == 1 (ap_helper.py:224-307: groundtruths2corners3d + parse_groundtruths);
    the corners of all B x MAX_NUM_OBJ labels are decoded in one batched float64 pass."""
    config = config_dict['dataset_config']
    center = end_points['center_label'][:, :, 0:3].float()
    size64 = config.mean_size(center.device).double()[end_points['size_class_label'].long()] + \
        end_points['size_residual_label'].double()
    heading64 = config.class2angle_f64(end_points['heading_class_label'].long(),
                                       end_points['heading_residual_label'].float())
    corners = corners_upright_camera(center, size64, heading64).cpu().numpy()
    mask = (end_points['box_label_mask'] == 1).cpu().numpy()
    sem = end_points['sem_cls_label'].cpu().numpy()
    batch = [[(int(sem[i, j]), corners[i, j]) for j in np.nonzero(mask[i])[0]]
             for i in range(mask.shape[0])]
    end_points['batch_gt_map_cls'] = batch
    return batch


class APCalculator(object):
    """ap_helper.py:382-435: accumulates per-scan predictions / ground truths, then one
    eval_det (votenet/eval_det.py) per compute_metrics with the oriented-box IoU."""

    def __init__(self, ap_iou_thresh=0.25, class2type_map=None, device=None):
        self.ap_iou_thresh = ap_iou_thresh
        self.class2type_map = class2type_map
        self.device = device
        self.reset()

    def step(self, batch_pred_map_cls, batch_gt_map_cls):
        bsize = len(batch_pred_map_cls)
        assert bsize == len(batch_gt_map_cls)
        for i in range(bsize):
            self.gt_map_cls[self.scan_cnt] = batch_gt_map_cls[i]
            self.pred_map_cls[self.scan_cnt] = batch_pred_map_cls[i]
            self.scan_cnt += 1

    def compute_metrics(self):
        from .eval_det import eval_det
        rec, prec, ap = eval_det(self.pred_map_cls, self.gt_map_cls, ovthresh=self.ap_iou_thresh,
                                 device=self.device)
        name = lambda key: self.class2type_map[key] if self.class2type_map else str(key)  # noqa: E731
        ret_dict = {}
        for key in sorted(ap.keys()):
            ret_dict['%s Average Precision' % name(key)] = ap[key]
        ret_dict['mAP'] = np.mean(list(ap.values()))
        rec_list = []
        for key in sorted(ap.keys()):
            last = rec[key][-1] if np.ndim(rec[key]) and len(rec[key]) else 0
            ret_dict['%s Recall' % name(key)] = last
            rec_list.append(last)
        ret_dict['AR'] = np.mean(rec_list)
        return ret_dict

    def reset(self):
        self.gt_map_cls = {}    # {scan_id: [(classname, bbox)]}
        self.pred_map_cls = {}  # {scan_id: [(classname, bbox, score)]}
        self.scan_cnt = 0
