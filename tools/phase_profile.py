"""Kernel count / device time per phase of the supervised step (torch.profiler), to see where
launch-bound work sits.  python tools/phase_profile.py"""
import importlib
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dioumatch_amd")
V = importlib.import_module("3dioumatch_amd.votenet")
from importlib import import_module
step_mod = import_module("3dioumatch_amd.votenet.step")
data_mod = import_module("3dioumatch_amd.votenet.data")
losses = import_module("3dioumatch_amd.votenet.losses")
config = import_module("3dioumatch_amd.votenet.config")

dev = torch.device("cuda:0")
cfg = config.scannet_config()
st = step_mod.SupervisedStep(cfg, dev, graphs=False)
batch = data_mod.make_batch(8, 40000, cfg, seed=1, device=dev)


def run(phase_out):
    b = dict(batch)
    st.optimizer.zero_grad(set_to_none=True)
    with phase_out("forward"):
        ep = st.model(b, mode="jitter")
    ep.update({k: v for k, v in b.items() if torch.is_tensor(v)})
    ep["all_supervised"] = True  # what SupervisedStep passes for a fully labeled batch
    with phase_out("loss"):
        loss, ep = losses.get_labeled_loss(ep, cfg, {"dataset_config": cfg})
    with phase_out("backward"):
        loss.backward()
    with phase_out("optimizer"):
        st.optimizer.step()


class Null:
    def __init__(self, n): pass
    def __enter__(self): return self
    def __exit__(self, *a): return False


for _ in range(3):
    run(Null)
torch.cuda.synchronize()
results = {}


class Phase:
    def __init__(self, name): self.name = name
    def __enter__(self):
        torch.cuda.synchronize()
        self.p = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
        self.p.__enter__()
    def __exit__(self, *a):
        torch.cuda.synchronize()
        self.p.__exit__(*a)
        results[self.name] = self.p
        return False


run(Phase)
for name, p in results.items():
    ev = [e for e in p.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    tot = sum(e.device_time for e in ev)
    small = [e for e in ev if e.device_time < 10]
    print("== %s: %d kernels, %.2f ms device; %d kernels <10us totalling %.2f ms"
          % (name, len(ev), tot / 1e3, len(small), sum(e.device_time for e in small) / 1e3))
    ka = p.key_averages()
    rows = sorted(ka, key=lambda k: -k.self_device_time_total)
    for k in rows[:10]:
        if k.self_device_time_total > 0:
            print("   %-60s n=%4d  %.3f ms" % (k.key[:60], k.count, k.self_device_time_total / 1e3))
    print("   -- host ops by launch count")
    ops = [k for k in ka if k.key.startswith("aten::") and k.self_device_time_total > 0]
    for k in sorted(ops, key=lambda k: -k.count)[:22]:
        print("   %-40s n=%4d  %.3f ms" % (k.key[:40], k.count, k.self_device_time_total / 1e3))
