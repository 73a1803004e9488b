"""Train-step harness: the body of the reference's training loops as reusable functions,
data-parallel over one node.

  supervised step   = pretrain.py:train_one_epoch body (pretrain.py:310-347):
                      forward_with_pred_jitter -> get_labeled_loss -> backward -> Adam
  EMA teacher       = train.py:update_ema_variables (train.py:285-289)
  schedules         = stair-step LR (pretrain.py:242-254 / train.py:242-254) and the
                      BN-momentum schedule (pretrain.py:217-221, pytorch_utils.BNMomentumScheduler)

Parallelism (new -- the reference only has a broken nn.DataParallel branch):
one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the
CPU tests), all parameters live in ONE flat buffer (1 063 985 fp32 = 4.26 MB) and so do the gradients, so
the data-parallel exchange is a single all-reduce per step between the backward graph and the
Adam graph (a 4 MB ring all-reduce is tens of microseconds on xGMI; there is nothing worth
overlapping).  BatchNorm buffers stay per replica like the reference's per-GPU BatchNorm (no
SyncBN).  Each rank draws its own scenes (weak scaling).  `wrap_ddp` remains for callers that
want torch's DistributedDataParallel around the module (the custom forward is reachable through
nn.Module.__call__, `model(batch, mode="jitter")`, so that DDP's hooks fire).
"""
import os
import weakref
import sys

import numpy as np
import torch

from pointnet2._mlp_ext import deferred_weight_reductions
from pointnet2.pytorch_utils import deferred_bn_counters, zero_grads_none

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


def flatten_parameters(module):
    """Re-home every parameter of `module` as a view into ONE contiguous fp32 buffer (in
    parameters() order) and return that buffer as an nn.Parameter.  Gradient exchange, the Adam
    update and the EMA teacher then each touch a single 4.26 MB tensor instead of 96 small ones.
    state_dict()/load_state_dict() keep working (they copy in place)."""
    params = list(module.parameters())
    total = sum(p.numel() for p in params)
    flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view(p.shape)
        off += n
    return torch.nn.Parameter(flat, requires_grad=True)


def _tensor_items(batch):
    return {k: v for k, v in batch.items() if torch.is_tensor(v)}


_SIDE_STREAMS = {}  # (device type, index) -> the prefetch stream shared by all runners


class SupervisedStep(object):
    """One optimisation step on a labeled batch (dict of tensors already on `device`).

    On the GPU the step is three HIP graphs, captured on the first call and replayed afterwards
    (about 1100 kernel launches per step, three quarters of them shorter than 10 us: replaying
    them removes the host from the critical path):

        G0  coordinate-only index chain (FPS x5, ball queries) of the NEXT batch, replayed on a
            side stream by prefetch_geometry() so it overlaps the dense kernels of this step
        G1  forward_with_pred_jitter -> get_labeled_loss -> backward -> gradients packed into
            one flat buffer
        --  world_size > 1: ONE all-reduce of the flat gradient (RCCL / gloo), between graphs
        G2  Adam on the flat parameter buffer

    In graph mode the returned loss / end_points are the graphs' STATIC output buffers: the next
    replay overwrites them, so callers that accumulate statistics across steps must clone().
    Inputs are staged into static buffers (`next` while G0 runs, copied to `cur` for G1).  The
    graphs bake in tensor shapes, the set of supervised samples (the contents of
    `supervised_mask`, see _mask_facts) and the BatchNorm momentum; a change of any of them
    re-captures.  On the CPU (tests, gloo) the same functions run eagerly.
    """

    def __init__(self, cfg, device, world_size=1, num_proposal=256, lr=1e-3, seed=0, graphs=None,
                 graphs_fallback=False):
        self.cfg = cfg
        # False: a failed HIP-graph capture raises.  True: it is reported on stderr and the
        # runner continues with eager launches (`runner.graphs` tells which one is running).
        self.graphs_fallback = bool(graphs_fallback)
        self.device = device
        self.world = world_size
        self.net = build_detector(cfg, num_proposal=num_proposal, seed=seed).to(device).train()
        self.model = self.net  # callers reach the custom forward through model(batch, mode=...)
        self._params = list(self.net.parameters())
        self.flat_params = flatten_parameters(self.net)
        self.flat_grad = torch.zeros_like(self.flat_params.data)
        self.flat_params.grad = self.flat_grad
        on_gpu = device.type == "cuda"
        self.graphs = on_gpu if graphs is None else (bool(graphs) and on_gpu)
        if os.environ.get("VOTENET_HIP_GRAPHS", "1") == "0":
            self.graphs = False
        lr_value = torch.tensor(float(lr), device=device) if on_gpu else lr
        # holder of the hyper-parameters and of the state (torch's keys: step, exp_avg,
        # exp_avg_sq); on the GPU the update itself is votenet_adam_step (see _apply)
        self.optimizer = torch.optim.Adam([self.flat_params], lr=lr_value, weight_decay=0,
                                          capturable=on_gpu)
        self._adam_scratch = None
        # the learning rate the Adam kernel reads: a device scalar that never moves, refreshed
        # from param_groups[0]["lr"] (tensor OR float) eagerly before every update, so a caller
        # that assigns a float -- the reference's adjust_learning_rate does -- is honoured under
        # graph replay too
        self._lr_scalar = torch.zeros((), dtype=torch.float32, device=device) if on_gpu else None
        self._side = None
        self._captured = None  # signature the graphs were captured for
        self._mask_cache = (None, None, None)
        self._token = 0
        self.global_step = 0

    # ---------------------------------------------------------------- schedules
    def set_epoch(self, epoch, base_lr=1e-3):
        for group in self.optimizer.param_groups:
            if torch.is_tensor(group["lr"]):
                group["lr"].fill_(lr_at(epoch, base_lr))
            else:
                group["lr"] = lr_at(epoch, base_lr)
        momentum = bn_momentum_at(epoch)
        for m in self.net.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                if m.momentum != momentum:
                    self._captured = None  # the momentum is a kernel argument baked into G1
                m.momentum = momentum

    # ---------------------------------------------------------------- checkpoint interchange
    def optimizer_state_dict(self):
        """The Adam state in the REFERENCE's layout -- torch.optim.Adam over net.parameters(), one
        entry per parameter tensor (pretrain.py:196,374 save / load `optimizer_state_dict`) --
        split out of the flat buffers, so checkpoints move both ways."""
        st = self.optimizer.state.get(self.flat_params, {})
        group = {k: (float(v) if torch.is_tensor(v) else v)
                 for k, v in self.optimizer.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self._params)))
        state, off = {}, 0
        for i, p in enumerate(self._params):
            n = p.numel()
            if st:
                step = st["step"]
                state[i] = {"step": step.detach().clone().float().cpu() if torch.is_tensor(step)
                            else torch.tensor(float(step)),
                            "exp_avg": st["exp_avg"][off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": st["exp_avg_sq"][off:off + n].view(p.shape).clone()}
            off += n
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, state_dict):
        """Inverse of optimizer_state_dict(): a per-parameter Adam state (e.g. from a reference
        checkpoint) merged into the flat moment buffers."""
        state = state_dict["state"]
        if not state:
            return
        if len(state) != len(self._params):
            raise ValueError("optimizer state holds %d tensors, the detector has %d parameters"
                             % (len(state), len(self._params)))
        flat = self.flat_params
        st = self.optimizer.state[flat]
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(flat.data)
            st["exp_avg_sq"] = torch.zeros_like(flat.data)
            st["step"] = torch.zeros((), dtype=torch.float32,
                                     device=flat.device if self.device.type == "cuda" else "cpu")
        off, steps = 0, []
        for i, p in enumerate(self._params):
            n = p.numel()
            entry = state[i] if i in state else state[str(i)]
            for k in ("exp_avg", "exp_avg_sq"):
                if tuple(entry[k].shape) != tuple(p.shape):
                    raise ValueError("optimizer state %d: %s has shape %s, the parameter %s"
                                     % (i, k, tuple(entry[k].shape), tuple(p.shape)))
            st["exp_avg"][off:off + n].copy_(entry["exp_avg"].reshape(-1))
            st["exp_avg_sq"][off:off + n].copy_(entry["exp_avg_sq"].reshape(-1))
            steps.append(float(entry["step"]))
            off += n
        if len(set(steps)) != 1:
            raise ValueError("per-parameter Adam steps differ: cannot merge into one flat state")
        if torch.is_tensor(st["step"]):
            st["step"].fill_(steps[0])
        else:
            st["step"] = steps[0]
        saved = state_dict["param_groups"][0]
        for k in ("betas", "eps", "weight_decay", "amsgrad"):
            if k in saved:
                self.optimizer.param_groups[0][k] = saved[k]
        if "lr" in saved:  # torch's optimizer.load_state_dict restores the rate as well
            cur = self.optimizer.param_groups[0]["lr"]
            if torch.is_tensor(cur):
                cur.fill_(float(saved["lr"]))
            else:
                self.optimizer.param_groups[0]["lr"] = float(saved["lr"])

    # ---------------------------------------------------------------- the step, eagerly
    # bf16 images of the convolution weights for the small layers' kernels, rebuilt by ONE launch at
    # the head of every forward pass (pointnet2/_mlp_ext.py WeightImages); STEP_WEIGHT_IMAGES=0: off
    use_weight_images = os.environ.get("STEP_WEIGHT_IMAGES", "1") != "0"

    def _images_of(self, net, slot):
        """the WeightImages of `net` (made on first use; None without a GPU)"""
        if not self.use_weight_images or self.device.type != "cuda":
            return None
        cache = self.__dict__.setdefault("_weight_images", {})
        if slot not in cache:
            from pointnet2 import _mlp_ext as K
            cache[slot] = K.WeightImages([p.data for p in net.parameters()
                                          if p.dim() >= 3 and p.shape[0] <= 512
                                          and p.numel() // p.shape[0] <= 512])
        return cache[slot]

    def _forward_backward(self, batch):
        from pointnet2 import _mlp_ext as K
        for p in self._params:
            p.grad = None
        images = self._images_of(self.net, "student")
        with K.weight_images(images):
            if images is not None:
                images.refresh()
            with deferred_bn_counters():
                end_points = self.model(batch, mode="jitter")
            end_points.update({k: v for k, v in batch.items()
                               if torch.is_tensor(v) or k in ("all_supervised", "labeled_num")})
            loss, end_points = get_labeled_loss(end_points, self.cfg, {"dataset_config": self.cfg})
            with zero_grads_none(), deferred_weight_reductions(self.defer_weight_reductions):
                loss.backward()
        self._pack_gradients()
        return loss, end_points

    # the ~30 partial-sum reductions that finish the weight gradients of a backward pass run as one
    # launch after it (nothing reads a weight gradient before _pack_gradients)
    defer_weight_reductions = os.environ.get("STEP_DEFER_WGRAD_REDUCE", "1") != "0"

    def _pack_gradients(self):
        """p.grad (fresh tensors from autograd) -> flat_grad, one concatenation kernel."""
        missing = [p for p in self._params if p.grad is None]
        if missing:
            # identically-zero gradients arrive as None (zero_grads_none): one cached zero buffer
            # stands in for all of them inside the same concatenation
            need = max(p.numel() for p in missing)
            zeros = getattr(self, "_grad_zeros", None)
            if zeros is None or zeros.numel() < need:
                zeros = self._grad_zeros = torch.zeros(need, dtype=self.flat_grad.dtype,
                                                       device=self.flat_grad.device)
            pieces = [p.grad.reshape(-1) if p.grad is not None else zeros[:p.numel()]
                      for p in self._params]
        else:
            pieces = [p.grad.reshape(-1) for p in self._params]
        torch.cat(pieces, out=self.flat_grad)

    # run the gradient collective even when world_size == 1 (a one-rank process group): lets a
    # single-GPU box exercise RCCL's init, the all-reduce and its ordering between the graphs
    exchange_always = False
    # record device time of every eager gradient all-reduce (events on the current stream) in
    # self.exchange_events: bench.py's N > 1 line reports their median
    time_exchange = False
    # capture the all-reduce at the head of the update graph G2 instead of launching it eagerly
    # between the two replays.  Opt-in (STEP_GRAPH_ALLREDUCE=1 or this attribute): a collective
    # inside a HIP graph can only be validated here with a ONE-rank nccl group (the box has one
    # GPU), so the default stays the eager launch that torch.distributed documents.
    capture_exchange = os.environ.get("STEP_GRAPH_ALLREDUCE") == "1"
    _exchange_in_graph = False

    def _exchanges(self):
        return self.world > 1 or self.exchange_always

    def _exchange_gradients(self):
        """Data parallelism: the mean of the per-rank gradients, one collective per step."""
        if not self._exchanges():
            return
        if self.time_exchange and self.device.type == "cuda" and \
                not torch.cuda.is_current_stream_capturing():
            if not hasattr(self, "exchange_events"):
                self.exchange_events = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.distributed.all_reduce(self.flat_grad)
            e1.record()
            self.exchange_events.append((e0, e1))
            return
        torch.distributed.all_reduce(self.flat_grad)

    def exchange_report(self):
        """{"backend", "world_size", "allreduce_us" (median device time, None if untimed),
        "bytes", "in_graph"} of the gradient collective."""
        us = None
        if getattr(self, "exchange_events", None):
            torch.cuda.synchronize(self.device)
            t = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in self.exchange_events)
            us = round(t[len(t) // 2], 2)
        dist = torch.distributed
        on = dist.is_available() and dist.is_initialized()
        return {"backend": dist.get_backend() if on else None,
                "world_size": dist.get_world_size() if on else 1,
                "allreduce_us": us, "samples": len(getattr(self, "exchange_events", [])),
                "bytes": int(self.flat_grad.numel() * self.flat_grad.element_size()),
                "in_graph": bool(self._exchange_in_graph)}

    def _apply(self, teacher=None, ema_weight=None):
        """Adam on the flat buffer (and, for the semi-supervised subclass, the teacher's EMA
        update): one kernel of the gfx950 library on the GPU, torch's optimizer elsewhere.  The
        state lives in self.optimizer.state under torch's own keys either way."""
        if self.device.type != "cuda":
            if self.world > 1:
                self.flat_grad.mul_(1.0 / self.world)
            self.optimizer.step()
            if teacher is not None:
                teacher.lerp_(self.flat_params.data, ema_weight)
            return
        import importlib
        _L = importlib.import_module("3dioumatch_amd._lib")
        group = self.optimizer.param_groups[0]
        st = self.optimizer.state[self.flat_params]
        if "exp_avg" not in st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=self.device)
            st["exp_avg"] = torch.zeros_like(self.flat_params.data)
            st["exp_avg_sq"] = torch.zeros_like(self.flat_params.data)
        if self._adam_scratch is None:
            self._adam_scratch = torch.zeros(2, dtype=torch.float32, device=self.device)
        lr = self._lr_scalar  # refreshed by _refresh_lr() before this call (never inside a capture)
        if group.get("amsgrad") or group.get("maximize"):
            raise RuntimeError("the flat Adam step implements neither amsgrad nor maximize")
        beta1, beta2 = group["betas"]
        with torch.cuda.device(self.device):
            _L.check(_L.lib.votenet_adam_step(
                self.flat_params.numel(), self.flat_params.data_ptr(), self.flat_grad.data_ptr(),
                st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(),
                lr.data_ptr(), float(beta1), float(beta2), float(group["eps"]),
                float(group["weight_decay"]), 1.0 / self.world,
                None if teacher is None else teacher.data_ptr(),
                None if teacher is None else ema_weight.data_ptr(), self._adam_scratch.data_ptr(),
                torch.cuda.current_stream(self.device).cuda_stream), "votenet_adam_step")

    def _expose_gradients(self):
        """Point every p.grad at its slice of the (averaged) flat gradient."""
        off = 0
        for p in self._params:
            p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)
            off += p.numel()

    # ---------------------------------------------------------------- hooks for subclasses
    def _compute_geometry(self, inputs):
        """Coordinate-only index chain of one batch (dict of tensors)."""
        return self.net.compute_geometry(inputs)

    def _host_info(self, src):
        """Host-side facts about the batch layout that the captured graphs bake in."""
        if "supervised_mask" not in src:  # a plain labeled batch
            return {"all_supervised": True}
        sup = torch.nonzero(src["supervised_mask"]).squeeze(1).long()
        self._supervised_inds = sup  # (kept on self: a graph input must outlive the capture)
        return {"supervised_inds": sup,
                "all_supervised": sup.numel() == src["supervised_mask"].numel()}

    def _state(self):
        """Every tensor a step mutates besides the optimizer state (restored after capture)."""
        return [b for b in self.net.buffers()] + [self.flat_params.data]

    def _refresh_lr(self):
        """param_groups[0]["lr"] -> the device scalar the (possibly captured) Adam kernel reads."""
        if self._lr_scalar is None:
            return
        lr = self.optimizer.param_groups[0]["lr"]
        if torch.is_tensor(lr):
            self._lr_scalar.copy_(lr, non_blocking=True)
        else:
            self._lr_scalar.fill_(float(lr))

    def _before_apply(self):
        """Eager host-side work between the backward graph and the update graph."""
        self.global_step += 1
        self._refresh_lr()

    # ---------------------------------------------------------------- geometry prefetch
    def side_stream(self):
        """The stream the prefetched index chains run on (None on CPU).  It has slack -- the chain
        is ~4 ms of a ~6 ms step -- so a feeding loop can put its host -> device copies on it too,
        instead of on a stream of its own that may land on the main stream's hardware queue."""
        if self.device.type != "cuda":
            return None
        if self._side is None:
            # ONE side stream per device for every runner of the process: HIP maps streams onto a
            # few hardware queues, and a third runner's fresh stream was seen to share the main
            # stream's queue (its index chain no longer overlapped the dense kernels: 16.8 ->
            # 27.3 ms per semi-supervised step inside bench.py's multi-workload run)
            key = (self.device.type, self.device.index)
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = torch.cuda.Stream(device=self.device)
            self._side = _SIDE_STREAMS[key]
        return self._side

    def prefetch_geometry(self, batch):
        """Launch the coordinate-only index computations (FPS chain, ball queries) of `batch`
        on a side stream; the step that later consumes `batch` waits for them.  Call it for
        batch i+1 right before running the step on batch i: the serial FPS rounds then overlap
        the dense kernels of step i instead of heading step i+1's critical path."""
        if self.device.type != "cuda":
            batch["geometry"] = self._compute_geometry(batch)
            return
        self.side_stream()
        main = torch.cuda.current_stream(self.device)
        if self.graphs and self._ensure_captured(batch):
            slot = self._slots[self._turn]
            self._turn ^= 1
            self._side.wait_stream(main)  # inputs are ready; the slot's last consumer is done
            with torch.cuda.stream(self._side):
                self._stage(slot, batch)
                slot["graph"].replay()
                slot["ready"].record(self._side)
            self._token += 1
            slot["token"] = self._token
            batch["geometry"] = slot["geometry"]
            batch["_staged"] = (slot, self._token)
            return
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            geometry = self._compute_geometry(batch)
            done = torch.cuda.Event()
            done.record(self._side)
        for t in geometry.values():
            if torch.is_tensor(t):
                t.record_stream(main)
        batch["geometry"] = geometry
        batch["_geometry_ready"] = done

    # ---------------------------------------------------------------- HIP graphs
    def _mask_facts(self, batch):
        """The CONTENTS of supervised_mask are host-side control flow baked into G1 (which samples
        the labeled loss indexes, how many are labeled), so they are part of the capture
        signature.  A loader should hand over a host copy (`supervised_mask_host`, any sequence of
        0/1) next to the device tensor; otherwise the device tensor is read back -- once per
        tensor object and in-place version, so a resident mask costs one synchronisation in total
        and a fresh mask per batch one per step."""
        mask = batch.get("supervised_mask")
        if mask is None:
            return None
        host = batch.get("supervised_mask_host")
        if host is None:
            # the same tensor OBJECT at the same in-place version (never its address: a fresh
            # mask per batch may land on a recycled one)
            ref, version, cached = self._mask_cache
            if ref is None or ref() is not mask or version != mask._version:
                cached = tuple(int(v) for v in mask.detach().cpu().reshape(-1).tolist())
                self._mask_cache = (weakref.ref(mask), mask._version, cached)
            host = cached
        return tuple(int(v) != 0 for v in host)

    def _signature(self, batch):
        shapes = tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in _tensor_items(batch).items()))
        return shapes, self._mask_facts(batch)

    def _stage(self, slot, batch):
        src = _tensor_items(batch)
        torch._foreach_copy_([slot["inputs"][k] for k in self._keys], [src[k] for k in self._keys])

    def _ensure_captured(self, batch):
        """Capture G0/G1/G2 for the shapes of `batch` (once).  Returns False -- and switches to
        the eager path for good -- if the capture is not possible on this stack."""
        sig = self._signature(batch)
        if self._captured == sig:
            return True
        try:
            self._capture(batch, sig)
            return True
        except Exception as err:  # noqa: BLE001
            if not self.graphs_fallback:
                raise  # a silent eager fallback would hide a broken capture (and a 1.6x slower step)
            sys.stderr.write("SupervisedStep: HIP graph capture failed (%s: %s); running the "
                             "step eagerly (graphs_fallback=True)\n" % (type(err).__name__, err))
            torch.cuda.synchronize(self.device)
            self.graphs = False
            self._captured = None
            return False

    def _capture(self, batch, sig):
        dev = self.device
        self._refresh_lr()  # the warm-up updates below read the device scalar
        torch.cuda.synchronize(dev)
        src = _tensor_items(batch)
        src.pop("supervised_inds", None)
        self._keys = sorted(src)
        # two staging slots (inputs + index chain of a prefetched batch) and the buffers G1 reads
        self._slots = [{"inputs": {k: src[k].clone(memory_format=torch.contiguous_format)
                                   for k in self._keys}, "token": -1,
                        "ready": torch.cuda.Event()} for _ in range(2)]
        self._turn = 0
        self._cur = {k: src[k].clone(memory_format=torch.contiguous_format) for k in self._keys}
        # which samples are supervised is part of the captured control flow (host-side nonzero)
        host_info = self._host_info(src)

        # state touched by the warm-up iterations and by the capture itself
        state = self._state()
        saved_state = [t.clone() for t in state]
        saved_rng = torch.cuda.get_rng_state(dev)
        had_state = len(self.optimizer.state) > 0
        saved_opt = {k: v.clone() for k, v in self.optimizer.state.get(self.flat_params, {}).items()
                     if torch.is_tensor(v)}

        def body_geometry(slot):
            return self._compute_geometry(slot["inputs"])

        def make_inputs():
            inputs = dict(self._cur)
            inputs["geometry"] = self._cur_geometry
            inputs.update(host_info)
            return inputs

        def body_step():
            return self._forward_backward(make_inputs())

        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(warm):
            for _ in range(2):  # lazy initialisations (MIOpen/hipBLASLt handles, autograd)
                geometry = body_geometry(self._slots[0])
                self._cur_geometry = {k: v.clone() for k, v in geometry.items()}
                body_step()
                self.flat_grad.zero_()
                self._apply()  # creates the Adam state; a zero gradient leaves the weights alone
        torch.cuda.current_stream(dev).wait_stream(warm)
        torch.cuda.synchronize(dev)

        mode = dict(capture_error_mode="thread_local")
        for slot in self._slots:
            slot["graph"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(slot["graph"], **mode):
                slot["geometry"] = body_geometry(slot)
        self._geo_keys = sorted(k for k, v in self._slots[0]["geometry"].items() if torch.is_tensor(v))
        self._cur_geometry = {k: self._slots[0]["geometry"][k].clone() for k in self._geo_keys}
        # Without a gradient exchange nothing separates the backward pass from the update: ONE graph
        # (the host-side learning-rate refresh of _before_apply moves in front of it -- only the
        # update kernel reads that scalar).  STEP_ONE_GRAPH=0: two graphs as with an exchange.
        self._merged = not self._exchanges() and os.environ.get("STEP_ONE_GRAPH", "1") != "0"
        self._capture_step(make_inputs, mode)
        self._exchange_in_graph = False
        if self.capture_exchange and self._exchanges() and torch.distributed.get_backend() == "nccl":
            try:  # the collective as the first node of G2 (the zero gradient makes it harmless here)
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, **mode):
                    torch.distributed.all_reduce(self.flat_grad)
                    self._apply()
                self._g2, self._exchange_in_graph = g2, True
            except Exception as err:  # noqa: BLE001
                sys.stderr.write("SupervisedStep: all-reduce not capturable (%s: %s); it stays an "
                                 "eager launch between the graphs\n" % (type(err).__name__, err))
                torch.cuda.synchronize(dev)
        if self._merged:
            self._g2 = None
        elif not self._exchange_in_graph:
            self._g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g2, **mode):
                self._apply()
        torch.cuda.synchronize(dev)

        # undo every side effect of warm-up and capture
        for t, saved in zip(state, saved_state):
            t.copy_(saved)
        for k, v in self.optimizer.state[self.flat_params].items():
            if torch.is_tensor(v):
                v.copy_(saved_opt[k]) if had_state else v.zero_()
        torch.cuda.set_rng_state(saved_rng, dev)
        self._expose_gradients()
        self._copy_dst = [self._cur[k] for k in self._keys] + \
            [self._cur_geometry[k] for k in self._geo_keys]
        for slot in self._slots:
            slot["copy_table"] = None  # pointer table of _stage_copies: rebuilt for the new buffers
            slot["copy_src"] = [slot["inputs"][k] for k in self._keys] + \
                [slot["geometry"][k] for k in self._geo_keys]
        torch.cuda.synchronize(dev)
        self._captured = sig

    def _capture_step(self, make_inputs, mode):
        """forward + loss + backward (+ the update when nothing separates them) as self._g1"""
        self._g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g1, **mode):
            self._loss, self._end_points = self._forward_backward(make_inputs())
            if self._merged:
                self._apply()

    def _replay_step(self):
        self._g1.replay()

    def _stage_copies(self, slot):
        """slot -> the buffers G1 reads: one launch for all ~46 tensors (a table of pointers that
        never change after capture), torch's per-tensor copies where there is no GPU."""
        if self.device.type != "cuda":
            torch._foreach_copy_(self._copy_dst, slot["copy_src"])
            return
        if slot.get("copy_table") is None:
            rows, biggest = [], 0
            for d, s_ in zip(self._copy_dst, slot["copy_src"]):
                if not (d.is_contiguous() and s_.is_contiguous() and d.dtype == s_.dtype
                        and d.numel() == s_.numel()):
                    raise RuntimeError("staging buffers must be contiguous twins")
                nbytes = d.numel() * d.element_size()
                rows.append([s_.data_ptr(), d.data_ptr(), nbytes])
                biggest = max(biggest, nbytes)
            slot["copy_table"] = torch.tensor(rows, dtype=torch.int64, device=self.device)
            slot["copy_biggest"] = biggest
        import importlib
        _L = importlib.import_module("3dioumatch_amd._lib")
        with torch.cuda.device(self.device):
            _L.check(_L.lib.pn2_multi_copy(len(self._copy_dst), slot["copy_table"].data_ptr(),
                                           slot["copy_biggest"],
                                           torch.cuda.current_stream(self.device).cuda_stream),
                     "pn2_multi_copy")

    def _replay(self, batch):
        main = torch.cuda.current_stream(self.device)
        slot, token = batch.pop("_staged", (None, None))
        if slot is not None and slot["token"] == token:
            main.wait_event(slot["ready"])
        else:  # not prefetched: stage and run the index chain inline
            slot = self._slots[self._turn]
            self._turn ^= 1
            if self._side is not None:
                main.wait_stream(self._side)  # a prefetch in flight may own the slot
            self._stage(slot, batch)
            slot["graph"].replay()
        slot["token"] = -1
        self._stage_copies(slot)
        if self._merged:
            self._before_apply()
            self._replay_step()
        else:
            self._replay_step()
            if not self._exchange_in_graph:
                self._exchange_gradients()
            self._before_apply()
            self._g2.replay()
        batch.pop("geometry", None)
        return self._loss, self._end_points

    # ---------------------------------------------------------------- entry point
    def __call__(self, batch):
        if self.graphs and self._ensure_captured(batch):
            return self._replay(batch)
        batch.pop("_staged", None)
        ready = batch.pop("_geometry_ready", None)
        if ready is not None:
            torch.cuda.current_stream(self.device).wait_event(ready)
        loss, end_points = self._forward_backward(batch)
        batch.pop("geometry", None)  # consumed: every step computes (or prefetches) its own
        self._exchange_gradients()
        self._before_apply()
        self._apply()
        self._expose_gradients()
        return loss, end_points


class SemiSupervisedStep(SupervisedStep):
    """One stage-2 (3DIoUMatch) step on a labeled+unlabeled batch (train.py:305-371):

        teacher (EMA) forward_with_pred_jitter on `ema_point_clouds`, no grad, train-mode BN
        student forward_with_pred_jitter on `point_clouds`
        loss = get_labeled_loss (labeled scenes) + 2.0 * get_unlabeled_loss (pseudo labels of the
               teacher, filtered by objectness / class / predicted IoU and the device-side LHS-NMS)
        backward -> [one all-reduce] -> Adam -> teacher <- a*teacher + (1-a)*student

    Same launch structure as SupervisedStep: the coordinate-only index chains of BOTH clouds of the
    next batch run one step ahead on a side stream (G0), forward/backward of both networks and all
    losses replay as one HIP graph (G1), Adam and the EMA update as another (G2).  The EMA weight
    a = min(1 - 1/(step+1), ema_decay) lives in a device scalar that is refreshed before G2.
    The teacher's parameters are a second flat buffer, so the EMA update is a single lerp."""

    def __init__(self, cfg, device, world_size=1, num_proposal=256, lr=2e-3, seed=0, graphs=None,
                 unlabeled_loss_weight=2.0, ema_decay=0.999, dataset="scannet", config_dict=None,
                 graphs_fallback=False):
        super().__init__(cfg, device, world_size=world_size, num_proposal=num_proposal, lr=lr,
                         seed=seed, graphs=graphs, graphs_fallback=graphs_fallback)
        from .losses_unlabeled import default_config_dict
        self.teacher = build_detector(cfg, num_proposal=num_proposal, seed=seed).to(device).train()
        for p in self.teacher.parameters():
            p.requires_grad_(False)  # train.py:176-178 (detach_)
        self.flat_teacher = flatten_parameters(self.teacher).data
        self.flat_teacher.copy_(self.flat_params.data)  # both start from the same checkpoint
        self.unlabeled_loss_weight = unlabeled_loss_weight
        self.ema_decay = ema_decay
        self.config_dict = config_dict or default_config_dict(cfg, dataset=dataset)
        self._ema_weight = torch.zeros((), device=device)  # 1 - a, read by the update graph
        self._teacher_stream = None
        self._teacher_replay = None
        self._gt = None

    # the BN-momentum schedule touches the student only (train.py:234-237): inherited set_epoch
    # walks self.net

    # The index chain depends on coordinates only -- not on either network's weights -- and its
    # sampling kernels run ONE workgroup per cloud for ~3 ms: the student's (labeled + unlabeled) and
    # the teacher's (unlabeled, other augmentation) clouds go through ONE chain of launches, side by
    # side on 20 CUs, instead of two chains back to back (5.6 ms of fps_bucket_rounds per step).
    batch_chains = os.environ.get("STEP_SEMI_ONE_CHAIN", "1") != "0"

    def _compute_geometry(self, inputs):
        pc, ema = inputs["point_clouds"], inputs["ema_point_clouds"]
        if self.batch_chains and pc.shape[1:] == ema.shape[1:] and pc.dtype == ema.dtype:
            nb = pc.shape[0]
            both = self.net.compute_geometry({"point_clouds": torch.cat([pc, ema], dim=0)})
            geometry = {}
            for k, v in both.items():
                if not torch.is_tensor(v) or v.dim() == 0 or v.shape[0] != nb + ema.shape[0]:
                    raise RuntimeError("geometry entry %r is not batched along dim 0" % k)
                geometry[k] = v[:nb]
                geometry["ema_" + k] = v[nb:]
            return geometry
        geometry = dict(self.net.compute_geometry(inputs))
        teacher = self.teacher.compute_geometry({"point_clouds": ema})
        geometry.update({"ema_" + k: v for k, v in teacher.items()})
        return geometry

    def _host_info(self, src):
        mask = src["supervised_mask"]
        labeled = int(torch.count_nonzero(mask))
        if not bool((mask[:labeled] != 0).all()):
            raise ValueError("labeled scenes must come first in the batch (train.py:321-325)")
        return {"labeled_num": labeled}

    def _state(self):
        return super()._state() + [b for b in self.teacher.buffers()] + [self.flat_teacher]

    # ---- the step in three pieces: the two forward passes meet only in the consistency loss ----
    def _geometries(self, batch):
        geometry = batch.get("geometry")
        if geometry is None:
            return None, None
        return ({k: v for k, v in geometry.items() if not k.startswith("ema_")},
                {k[4:]: v for k, v in geometry.items() if k.startswith("ema_")})

    def _teacher_forward(self, batch):
        from pointnet2 import _mlp_ext as K
        images = self._images_of(self.teacher, "teacher")
        with K.weight_images(images), deferred_bn_counters(), torch.no_grad():
            if images is not None:
                images.refresh()
            return self.teacher({"point_clouds": batch["ema_point_clouds"],
                                 "geometry": self._geometries(batch)[1],
                                 "jitter_noise": batch.get("_teacher_noise")}, mode="jitter")

    def _student_forward(self, batch):
        from pointnet2 import _mlp_ext as K
        for p in self._params:
            p.grad = None
        images = self._images_of(self.net, "student")
        with K.weight_images(images), deferred_bn_counters():
            if images is not None:
                images.refresh()
            return self.model({"point_clouds": batch["point_clouds"],
                               "geometry": self._geometries(batch)[0],
                               "jitter_noise": batch.get("_student_noise")}, mode="jitter")

    def _losses_backward(self, end_points, ema_end_points, batch):
        from .losses_unlabeled import get_unlabeled_loss
        end_points.update({k: v for k, v in batch.items()
                           if torch.is_tensor(v) and k not in ("point_clouds", "ema_point_clouds")})
        labeled = batch.get("labeled_num")
        if labeled is None:
            labeled = self._host_info(batch)["labeled_num"]
        end_points["labeled_num"] = labeled
        from . import fused_loss
        if fused_loss.semi_loss_supported(end_points, labeled):
            # both losses as ONE autograd node over one gradient buffer per head output
            # (fused_loss._FusedSemiLoss); the pseudo labels first, without their loss
            _, end_points = get_unlabeled_loss(end_points, ema_end_points, self.cfg, self.config_dict,
                                               labels_only=True)
            loss, end_points = fused_loss.get_semi_loss_fused(end_points, self.cfg, labeled,
                                                              self.unlabeled_loss_weight)
        else:
            detection_loss, end_points = get_labeled_loss(end_points, self.cfg,
                                                          {"dataset_config": self.cfg})
            unlabeled_loss, end_points = get_unlabeled_loss(end_points, ema_end_points, self.cfg,
                                                            self.config_dict)
            loss = detection_loss + unlabeled_loss * self.unlabeled_loss_weight
            end_points["loss"] = loss
        from pointnet2 import _mlp_ext as K
        # (the student's images were rebuilt at the head of its forward pass)
        with K.weight_images(self._images_of(self.net, "student")), zero_grads_none(), \
                deferred_weight_reductions(self.defer_weight_reductions):
            loss.backward()
        self._pack_gradients()
        return loss, end_points

    def _forward_backward(self, batch):
        ema_end_points = self._teacher_forward(batch)
        end_points = self._student_forward(batch)
        return self._losses_backward(end_points, ema_end_points, batch)

    # Captured, the teacher's pass is a graph of its own, replayed on its own stream next to the
    # student's forward graph: two chains of mostly small kernels fill each other's gaps (5.5 ->
    # 4.5 ms for the two forward passes, tools/two_forward_graphs.py).  As two separately launched
    # graphs -- as parallel branches of ONE captured graph the same kernels took 3.5 ms LONGER
    # (profiles/r5_step_experiments.json).  STEP_SEMI_SPLIT_GRAPHS=0: one graph, one stream.
    split_graphs = os.environ.get("STEP_SEMI_SPLIT_GRAPHS", "1") != "0"

    def _capture_step(self, make_inputs, mode):
        if not self.split_graphs:
            self._gt = None
            return super()._capture_step(make_inputs, mode)
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        if self._teacher_stream is None:
            self._teacher_stream = self._another_stream(cur, self._side)
        ts = self._teacher_stream
        # No random draw inside the two forward graphs: torch keeps ONE device-side (seed, offset)
        # pair per generator and every graph replay fills it on the replaying stream -- of two
        # graphs replayed side by side, both draw from whichever fill landed last (the student drew
        # its box jitter from the teacher's offset in one run of three with another process on the
        # GPU, the teacher from the student's otherwise; profiles/r5_semi_step_branches.txt).  The
        # four noise tensors are drawn on the main stream ahead of the fork, in the eager step's
        # order (teacher's two, student's two: the same numbers), into buffers the graphs read.
        probe = make_inputs()
        k = self.net.num_proposal
        shape = lambda key: (probe[key].shape[0], k, 3)  # noqa: E731
        self._noise = {name: tuple(torch.empty(shape(key), dtype=torch.float32, device=dev)
                                   for _ in range(2))
                       for name, key in (("_teacher_noise", "ema_point_clouds"),
                                         ("_student_noise", "point_clouds"))}
        self._draw_noise()
        plain_inputs = make_inputs

        def make_inputs():
            inputs = plain_inputs()
            inputs.update(self._noise)
            return inputs

        # once eagerly on that stream (whatever a module creates on first use -- its ticket
        # counters, _mlp_ext.tickets_of -- exists before the capture; the kernel library itself
        # keeps no state per stream: include/mlp_hip.h, `tickets`)
        ts.wait_stream(cur)
        with torch.cuda.stream(ts):
            self._teacher_forward(make_inputs())
        cur.wait_stream(ts)
        torch.cuda.synchronize(dev)
        self._gt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._gt, stream=ts, **mode):
            self._ema_end_points = self._teacher_forward(make_inputs())
        self._g1a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g1a, **mode):
            inputs = make_inputs()
            student = self._student_forward(inputs)
        self._g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g1, pool=self._g1a.pool(), **mode):
            self._loss, self._end_points = self._losses_backward(student, self._ema_end_points, inputs)
            if self._merged:
                self._apply()
        self._pick_teacher_replay_stream()

    def _another_stream(self, *others):
        """A stream whose HANDLE is none of `others`' (torch.cuda.Stream() hands out the streams of
        a pool of 32 round-robin).  Performance only: a teacher stream that IS the main or the
        prefetch stream would run the teacher's graph behind their work instead of beside it; any
        stream gives the same numbers (the modules own their ticket counters)."""
        taken = {s.cuda_stream for s in others if s is not None}
        stream = torch.cuda.Stream(device=self.device)
        for _ in range(64):
            if stream.cuda_stream not in taken:
                break
            stream = torch.cuda.Stream(device=self.device)
        return stream

    def _pick_teacher_replay_stream(self):
        """HIP maps streams onto a few hardware queues in creation order; a stream that shares the
        main stream's queue runs the teacher's graph AFTER the student's instead of beside it (seen
        when this runner is the third of a process: 10.9 instead of 9.8 ms per step).  A graph can be
        replayed on any stream, so the pair of forward graphs is timed on a few and the fastest
        keeps the job (the replays' side effects are put back here, whatever happens)."""
        dev = self.device
        self._teacher_replay = self._teacher_stream
        if os.environ.get("STEP_SEMI_TEACHER_PROBE", "1") == "0":
            return
        cur = torch.cuda.current_stream(dev)
        side = self.side_stream()
        candidates = [self._teacher_stream]
        for _ in range(3):  # distinct handles, none of them the main or the prefetch stream
            candidates.append(self._another_stream(cur, side, *candidates))
        times = []
        # the probe's replays move the BatchNorm buffers: put back whatever happens
        state = self._state()
        saved = [t.clone() for t in state]
        try:
            for s in candidates:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for rep in range(3):
                    if rep == 1:
                        torch.cuda.synchronize(dev)
                        t0.record(cur)
                    # as in the running loop: the next batch's index chain (4 ms of serial sampling
                    # rounds) is in flight on the prefetch stream -- a candidate that shares ITS queue
                    # would run the teacher behind it
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        self._slots[0]["graph"].replay()
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        self._gt.replay()
                    self._g1a.replay()
                    cur.wait_stream(s)
                t1.record(cur)
                torch.cuda.synchronize(dev)
                times.append(t0.elapsed_time(t1) / 2)
        finally:
            torch.cuda.synchronize(dev)
            for t, v in zip(state, saved):
                t.copy_(v)
        self._teacher_probe_ms = times
        self._teacher_replay = candidates[times.index(min(times))]
        self._teacher_probe_pick = times.index(min(times))
        if os.environ.get("STEP_DEBUG"):
            sys.stderr.write("teacher replay stream probe (ms per pair of forward graphs): %s\n"
                             % ", ".join("%.3f" % t for t in times))

    def _draw_noise(self):
        for pair in (self._noise["_teacher_noise"], self._noise["_student_noise"]):
            for t in pair:
                t.normal_()

    def _replay_step(self):
        if self._gt is None:
            return super()._replay_step()
        main = torch.cuda.current_stream(self.device)
        self._draw_noise()
        ts = self._teacher_replay
        ts.wait_stream(main)  # the inputs are staged
        with torch.cuda.stream(ts):
            self._gt.replay()
        self._g1a.replay()
        main.wait_stream(ts)
        self._g1.replay()

    def _before_apply(self):
        self.global_step += 1
        self._refresh_lr()
        a = min(1 - 1 / (self.global_step + 1), self.ema_decay)  # train.py:285-289
        self._ema_weight.fill_(1 - a)

    def _apply(self):
        super()._apply(self.flat_teacher, self._ema_weight)


def flat_grads(module):
    return torch.cat([p.grad.reshape(-1) for p in module.parameters() if p.grad is not None])


def flat_params(module):
    return torch.cat([p.detach().reshape(-1) for p in module.parameters()])


def shift_invariant_parameter_names(net):
    """Names of the parameters whose gradient is mathematically zero wherever the pooled maxima are
    positive: the bias of the LAST BatchNorm of every max-pooling set-abstraction module.  A
    constant shift of a pooled channel survives max / interpolation / concatenation unchanged and
    is removed by the batch-statistics BatchNorm behind the next 1x1 convolution, so what the
    backward pass computes for it is round-off of either sign -- which Adam's first steps turn into
    +-lr moves (profiles/r4_step_repeatability.txt).  Multi-step equivalence tests freeze them
    (freeze_shift_invariant_parameters) so that the rest can be held to a tight bound."""
    names = []
    for mod_name, mod in net.named_modules():
        mlp = getattr(mod, "mlp_module", None)
        if mlp is None or getattr(mod, "pooling", None) != "max" or len(mlp) == 0:
            continue
        last = len(mlp) - 1
        names.append("%s.mlp_module.layer%d.bn.bn.bias" % (mod_name, last))
    have = dict(net.named_parameters())
    return [n for n in names if n in have]


def freeze_shift_invariant_parameters(net):
    """requires_grad_(False) on shift_invariant_parameter_names(net): autograd then returns no
    gradient for them, the gradient packing substitutes zeros and Adam leaves them where they are.
    Call before the first step of a runner (the HIP graphs bake the set in).  -> the names."""
    names = shift_invariant_parameter_names(net)
    have = dict(net.named_parameters())
    for n in names:
        have[n].requires_grad_(False)
    return names
