"""Experiment: step time with different dense-layer back ends (not part of the product)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
V = importlib.import_module("3dioumatch_amd.votenet")
data = importlib.import_module("3dioumatch_amd.votenet.data")
mode = sys.argv[1] if len(sys.argv) > 1 else "base"
if mode == "nocudnn":
    torch.backends.cudnn.enabled = False
if mode in ("matmul", "matmul_nocudnn"):
    import torch.nn as nn
    def conv2d_fwd(self, x):
        if self.kernel_size == (1, 1) and self.bias is None:
            b, c, h, w = x.shape
            y = torch.matmul(self.weight.view(self.out_channels, c), x.reshape(b, c, h * w))
            return y.view(b, self.out_channels, h, w)
        return nn.functional.conv2d(x, self.weight, self.bias)
    nn.Conv2d.forward = conv2d_fwd
    if mode == "matmul_nocudnn":
        torch.backends.cudnn.enabled = False
dev = torch.device("cuda:0")
cfg = V.scannet_config()
runner = V.SupervisedStep(cfg, dev, num_proposal=256)
batch = data.make_batch(8, 40000, cfg, seed=100, device=dev)
for _ in range(4):
    runner(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    runner(batch)
torch.cuda.synchronize()
print(mode, "ms/step", (time.perf_counter() - t0) / 8 * 1e3)
