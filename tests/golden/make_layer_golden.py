"""Regenerate tests/golden/layers_ref.npz -- run in the BUILD container only (imports the
reference's Python from /root/reference; nothing of it is stored, only inputs/outputs).

The REFERENCE layers QueryAndGroup (pointnet2_utils.py:295-377), PointnetSAModuleVotes
(pointnet2_modules.py:169-277), PointnetFPModule (:362-422) and Pointnet2Backbone
(models/backbone_module.py:24-133) are executed on the CPU with `pointnet2._ext` supplied by
the oracle (oracle/standin.py; the real extension is CUDA-only).  Weights are generated from
a numpy seed (function `seeded_state` below, shared with tests/test_layers.py), so only the
seed travels.  Backward results use the TRUE three_interpolate gradient.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, ROOT)
sys.path.insert(0, TESTS)
from oracle.oracle import Oracle  # noqa: E402
from oracle import standin as oracle_ext  # noqa: E402

REF = "/root/reference"


def seeded_state(module, seed):
    """Deterministic parameters/buffers for `module`, independent of torch's RNG."""
    g = np.random.default_rng(seed)
    sd = module.state_dict()
    out = {}
    for k in sorted(sd):
        v = sd[k]
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy(g.uniform(0.5, 1.5, tuple(v.shape)).astype(np.float32))
        elif k.endswith("running_mean") or k.endswith("bias"):
            out[k] = torch.from_numpy((g.standard_normal(tuple(v.shape)) * 0.1).astype(np.float32))
        elif k.endswith("bn.weight"):
            out[k] = torch.from_numpy(g.uniform(0.8, 1.2, tuple(v.shape)).astype(np.float32))
        else:
            fan_in = int(np.prod(v.shape[1:])) if v.dim() > 1 else 1
            out[k] = torch.from_numpy(
                (g.standard_normal(tuple(v.shape)) * np.sqrt(2.0 / fan_in)).astype(np.float32))
    module.load_state_dict(out)
    return module


def sub(x):
    """strided sample of a big tensor, enough to pin it"""
    x = x.detach().numpy()
    return np.ascontiguousarray(x[:, ::4, ::4])


def main():
    o = Oracle()
    # --- import the reference with the oracle-backed extension
    ext = oracle_ext.make(o)
    ext.__name__ = "pointnet2._ext"
    sys.modules["pointnet2._ext"] = ext
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "pointnet2"))
    import pointnet2.pointnet2_utils as ref_utils  # noqa: E402
    import pointnet2_modules as ref_modules  # noqa: E402  (flat, as the reference models do)
    sys.path.insert(0, os.path.join(REF, "models"))
    from backbone_module import Pointnet2Backbone  # noqa: E402

    synth_spec = __import__("importlib").import_module("3dioumatch_amd.synth")
    out = {}
    g = np.random.default_rng(0)

    # 1. QueryAndGroup
    xyz = torch.from_numpy(synth_spec.cloud_uniform(2, 300, 1.0, seed=1))
    feats = torch.from_numpy(g.standard_normal((2, 4, 300)).astype(np.float32))
    new_xyz = xyz[:, :40].contiguous()
    for norm in (0, 1):
        qg = ref_utils.QueryAndGroup(0.3, 8, use_xyz=True, ret_grouped_xyz=True,
                                     normalize_xyz=bool(norm))
        nf, gx = qg(xyz, new_xyz, feats)
        out["qg%d_new_features" % norm] = nf.numpy()
        out["qg%d_grouped_xyz" % norm] = gx.numpy()
    out.update(qg_xyz=xyz.numpy(), qg_feats=feats.numpy())

    # 2. PointnetSAModuleVotes, train- and eval-mode BN, forward + backward
    sa = seeded_state(ref_modules.PointnetSAModuleVotes(npoint=64, radius=0.3, nsample=8,
                                                        mlp=[4, 8, 16], use_xyz=True,
                                                        normalize_xyz=True), seed=11)
    for mode in ("train", "eval"):
        sa.train(mode == "train")
        f = feats.clone().requires_grad_(True)
        x = xyz.clone().requires_grad_(True)
        nx, nf, inds = sa(x, f)
        w = torch.from_numpy(g.standard_normal(tuple(nf.shape)).astype(np.float32))
        (nf * w).sum().backward()
        out.update({"sa_%s_new_xyz" % mode: nx.detach().numpy(),
                    "sa_%s_new_features" % mode: nf.detach().numpy(),
                    "sa_%s_inds" % mode: inds.numpy(), "sa_%s_w" % mode: w.numpy(),
                    "sa_%s_grad_feats" % mode: f.grad.numpy(),
                    "sa_%s_grad_xyz" % mode: x.grad.numpy(),
                    "sa_%s_grad_conv0" % mode:
                        sa.mlp_module.layer0.conv.weight.grad.detach().numpy().copy()})
        sa.zero_grad()
    # given inds (vote aggregation style, proposal_module.py:105-106)
    sa.eval()
    given = torch.from_numpy(g.permutation(300)[:64].astype(np.int32)).repeat(2, 1).contiguous()
    nx, nf, inds = sa(xyz, feats, given)
    out.update(sa_given_inds=given.numpy(), sa_given_new_features=nf.detach().numpy())

    # 3. PointnetFPModule
    fp = seeded_state(ref_modules.PointnetFPModule(mlp=[16 + 4, 8]), seed=12).train()
    known = new_xyz
    known_feats = torch.from_numpy(g.standard_normal((2, 16, 40)).astype(np.float32)).requires_grad_(True)
    unk_feats = feats.clone().requires_grad_(True)
    y = fp(xyz, known, unk_feats, known_feats)
    w = torch.from_numpy(g.standard_normal(tuple(y.shape)).astype(np.float32))
    (y * w).sum().backward()
    out.update(fp_known_feats=known_feats.detach().numpy(), fp_out=y.detach().numpy(),
               fp_w=w.numpy(), fp_grad_known=known_feats.grad.numpy(),
               fp_grad_unknown=unk_feats.grad.numpy())

    # 4. the whole backbone (fixed hyper-parameters of the reference), B=1, N=4096, eval + train
    bb = seeded_state(Pointnet2Backbone(input_feature_dim=1), seed=13)
    pc = synth_spec.cloud_uniform(1, 4096, synth_spec.cube_side(4096, 0.2, 16), seed=5)
    height = pc[..., 2:3] - pc[..., 2].min()
    pc = torch.from_numpy(np.concatenate([pc, height], axis=2).astype(np.float32))
    out["bb_pc"] = pc.numpy()
    for mode in ("eval", "train"):
        bb.train(mode == "train")
        with torch.no_grad():
            ep = bb(pc)
        out["bb_%s_sa1_inds" % mode] = ep["sa1_inds"].numpy()
        out["bb_%s_sa2_inds" % mode] = ep["sa2_inds"].numpy()
        out["bb_%s_fp2_inds" % mode] = ep["fp2_inds"].numpy()
        for k in ("sa1_features", "sa2_features", "sa4_features", "fp2_features"):
            out["bb_%s_%s" % (mode, k)] = sub(ep[k])
        out["bb_%s_sa4_xyz" % mode] = ep["sa4_xyz"].numpy()
    path = os.path.join(HERE, "layers_ref.npz")
    np.savez_compressed(path, **out)
    print("layers_ref.npz %.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
