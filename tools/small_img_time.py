"""us per launch of the small layers' forward (weights from their bf16 images, as the step runs them): 16
back-to-back launches per graph replay, for A/B of library variants (PN2_LIB_SUFFIX).
    python tools/small_img_time.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
K = importlib.import_module("pointnet2._mlp_ext")
import bench
dev = torch.device("cuda:0")
B = 8
out = []
for name, m, k, r in (("fp2_l1", 256, 512, 1024), ("fp2_l2", 256, 256, 1024), ("fp1_l1", 256, 512, 512), ("prop_l1", 128, 128, 256)):
    w = torch.randn(m, k, device=dev) / k ** 0.5
    x = torch.randn(B, k, r, device=dev)
    ck = (torch.rand(k, device=dev) + 0.5, torch.rand(k, device=dev) + 0.5)
    imgs = K.WeightImages([w])
    imgs.refresh()
    with K.weight_images(imgs):
        def run():
            for _ in range(16):
                K.gemm_forward(w, x, ck)
        us = bench.time_op(run, iters=3, warm=2) / 16
    out.append("%s %.1f" % (name, us))
print("lib[%s] small forward with images, us per launch: %s" % (os.environ.get("PN2_LIB_SUFFIX", ""), " | ".join(out)))
