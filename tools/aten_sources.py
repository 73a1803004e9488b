"""Which Python lines issue the tensor (aten) operators of one supervised train step?  A
TorchDispatchMode records every aten call on CUDA tensors with the innermost frame inside this
repository (forward, loss and the Python bodies of the custom autograd functions; the mode is
re-entered in the autograd worker thread through a hook on the engine's first function).
    python tools/aten_sources.py [semi]"""
import collections
import importlib
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
V = importlib.import_module("3dioumatch_amd.votenet")
data = importlib.import_module("3dioumatch_amd.votenet.data")
dev = torch.device("cuda:0")
cfg = V.scannet_config()
SEMI = len(sys.argv) > 1 and sys.argv[1] == "semi"
if SEMI:
    runner = V.SemiSupervisedStep(cfg, dev, world_size=1, num_proposal=256, graphs=False)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v)
             for k, v in data.make_semi_batch(4, 8, 40000, cfg, seed=100).items()}
else:
    runner = V.SupervisedStep(cfg, dev, world_size=1, num_proposal=256, graphs=False)
    batch = data.make_batch(8, 40000, cfg, seed=100, device=dev)
SKIP = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::transpose", "aten::permute", "aten::slice",
        "aten::select", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias",
        "aten::as_strided", "aten::t", "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::size",
        "aten::stride", "aten::is_contiguous", "aten::unbind", "aten::split", "aten::_local_scalar_dense",
        "aten::lift_fresh", "aten::narrow", "aten::view_as", "aten::numel", "aten::to",
        "aten::_to_copy", "aten::item", "aten::sym_size", "aten::unflatten", "aten::flatten", "aten::chunk")
agg = collections.Counter()
size = collections.Counter()  # elements of the largest tensor argument, summed per (op, line)


class Tracer(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        cuda = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
        if cuda and name not in SKIP:
            frame = "?"
            for fs in reversed(traceback.extract_stack()[:-1]):
                fn = fs.filename
                if ("3dioumatch_amd" in fn) and "tools/" not in fn:
                    frame = "%s:%d %s" % (fn.split("3dioumatch_amd/")[-1], fs.lineno, fs.name)
                    break
            agg[(name, frame)] += 1
            size[(name, frame)] += max([a.numel() for a in list(args) + list((kwargs or {}).values())
                                        if isinstance(a, torch.Tensor)] or [0])
        return func(*args, **(kwargs or {}))


inputs = dict(batch)
inputs.update(runner._host_info(batch))
for _ in range(2):
    runner._forward_backward(dict(inputs))
torch.cuda.synchronize()
with Tracer():
    runner._forward_backward(dict(inputs))
torch.cuda.synchronize()
print("aten calls on GPU tensors in one forward + loss + backward (views / allocations skipped): %d" % sum(agg.values()))
for (name, frame), n in agg.most_common(200):
    print("%4d  %-26s %10d  %s" % (n, name, size[(name, frame)], frame))
