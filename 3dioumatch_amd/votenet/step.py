"""Train-step harness: the body of the reference's training loops as reusable functions,
data-parallel over one node.

  supervised step   = pretrain.py:train_one_epoch body (pretrain.py:310-347):
                      forward_with_pred_jitter -> get_labeled_loss -> backward -> Adam
  EMA teacher       = train.py:update_ema_variables (train.py:285-289)
  schedules         = stair-step LR (pretrain.py:242-254 / train.py:242-254) and the
                      BN-momentum schedule (pretrain.py:217-221, pytorch_utils.BNMomentumScheduler)

Parallelism (new -- the reference only has a broken nn.DataParallel branch):
one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the
CPU tests), DistributedDataParallel with ONE gradient bucket (1 063 985 fp32 = 4.26 MB, so a
single all-reduce per step that overlaps the tail of backward), per-replica BatchNorm buffers
like the reference's per-GPU BatchNorm (no SyncBN; broadcast_buffers=False).  Each rank draws
its own scenes (weak scaling).  The custom forward is reached through nn.Module.__call__
(`model(batch, mode="jitter")`) so that DDP's hooks fire.
"""
import numpy as np
import torch

from .detector import VoteNet
from .losses import get_labeled_loss


def build_detector(cfg, num_proposal=256, input_feature_dim=1, sampling="seed_fps", seed=0):
    torch.manual_seed(seed)  # identical initial weights on every rank
    return VoteNet(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster, cfg.mean_size_arr,
                   cfg, input_feature_dim=input_feature_dim, num_proposal=num_proposal,
                   sampling=sampling)


def wrap_ddp(net, device, world_size):
    if world_size <= 1:
        return net
    kwargs = dict(broadcast_buffers=False, bucket_cap_mb=32, gradient_as_bucket_view=True)
    if device.type == "cuda":
        kwargs["device_ids"] = [device.index]
    return torch.nn.parallel.DistributedDataParallel(net, **kwargs)


def lr_at(epoch, base_lr=1e-3, decay_steps=(400, 600, 800), decay_rates=(0.1, 0.1, 0.1)):
    """Stair-step learning rate (pretrain.py:242-248)."""
    lr = base_lr
    for step, rate in zip(decay_steps, decay_rates):
        if epoch >= step:
            lr *= rate
    return lr


def bn_momentum_at(epoch, init=0.5, decay_rate=0.5, decay_interval=20):
    """max(0.5 * 0.5^(epoch // 20), 0.001) (pretrain.py:217-219)."""
    return max(init * decay_rate ** (int(epoch / decay_interval)), 0.001)


@torch.no_grad()
def update_ema_variables(model, ema_model, alpha, global_step):
    """theta_T <- a*theta_T + (1-a)*theta_S over parameters() with
    a = min(1 - 1/(step+1), alpha) (train.py:285-289).  One fused multi-tensor update instead of
    the reference's 96 per-tensor mul_/add_ pairs."""
    a = min(1 - 1 / (global_step + 1), alpha)
    ema_params = [p.data for p in ema_model.parameters()]
    params = [p.data for p in model.parameters()]
    torch._foreach_mul_(ema_params, a)
    torch._foreach_add_(ema_params, params, alpha=1 - a)


class SupervisedStep(object):
    """One optimisation step on a labeled batch (dict of tensors already on `device`)."""

    def __init__(self, cfg, device, world_size=1, num_proposal=256, lr=1e-3, seed=0):
        self.cfg = cfg
        self.device = device
        self.net = build_detector(cfg, num_proposal=num_proposal, seed=seed).to(device).train()
        self.model = wrap_ddp(self.net, device, world_size)
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr, weight_decay=0)
        self._side = None

    def set_epoch(self, epoch, base_lr=1e-3):
        for group in self.optimizer.param_groups:
            group["lr"] = lr_at(epoch, base_lr)
        momentum = bn_momentum_at(epoch)
        for m in self.net.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                m.momentum = momentum

    def prefetch_geometry(self, batch):
        """Launch the coordinate-only index computations (FPS chain) of `batch` on a side
        stream; the step that later consumes `batch` waits for them.  Call it for batch i+1
        right before running the step on batch i: the serial FPS rounds then overlap the
        dense kernels of step i instead of heading step i+1's critical path."""
        if self.device.type != "cuda":
            batch["geometry"] = self.net.compute_geometry(batch)
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)  # inputs produced on the main stream are ready
        with torch.cuda.stream(self._side):
            geometry = self.net.compute_geometry(batch)
            done = torch.cuda.Event()
            done.record(self._side)
        for t in geometry.values():
            if torch.is_tensor(t):
                t.record_stream(main)
        batch["geometry"] = geometry
        batch["_geometry_ready"] = done

    def __call__(self, batch):
        ready = batch.pop("_geometry_ready", None)
        if ready is not None:
            torch.cuda.current_stream(self.device).wait_event(ready)
        self.optimizer.zero_grad(set_to_none=True)
        end_points = self.model(batch, mode="jitter")
        batch.pop("geometry", None)  # consumed: every step computes (or prefetches) its own
        end_points.update({k: v for k, v in batch.items() if torch.is_tensor(v)})
        loss, end_points = get_labeled_loss(end_points, self.cfg, {"dataset_config": self.cfg})
        loss.backward()
        self.optimizer.step()
        return loss, end_points


def flat_grads(module):
    return torch.cat([p.grad.reshape(-1) for p in module.parameters() if p.grad is not None])


def flat_params(module):
    return torch.cat([p.detach().reshape(-1) for p in module.parameters()])
