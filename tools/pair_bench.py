"""The north-star kernel pair at B=8, N=40000, m=2048, nsample=64, r=0.2 -- alone.

Two forms of the same work (38 516 736 algorithmic bytes):
  api   : ball_query + group_points(xyz, C=3) + group_points(feat, C=1), the three calls of the
          reference's operator surface;
  fused : query_and_group, self-contained: cell-list build (two kernels), then ONE kernel that
          answers the queries and writes the (B, 4, m, ns) tensor;
  layer : the same ONE kernel on the cell lists the layer's furthest-point-sampling kernel left
          behind (how PointnetSAModuleVotes runs SA1: no build between sampling and grouping).

    python tools/pair_bench.py [iters] [--json out.json]        # events around HIP-graph replays
    rocprofv3 --kernel-trace --stats ... -- python tools/pair_bench.py 20
    rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python tools/pair_bench.py 5 --plain
    rocprofv3 --kernel-trace --pmc WRITE_SIZE ... -- python tools/pair_bench.py 5 --plain
(--plain: eager launches only, no graph capture -- what the counter passes want;
 --only layer|fused|api: that form alone, so that a rocprofv3 stats / counter file holds ONE form
 per kernel name -- `fused` and `layer` launch the same kernel.)
"""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
ext = importlib.import_module("pointnet2._ext")
synth = importlib.import_module("3dioumatch_amd.synth")
import bench  # noqa: E402  (time_op: events around a HIP-graph replay)

args = [a for a in sys.argv[1:] if not a.startswith("--")]
iters = int(args[0]) if args else 20
plain = "--plain" in sys.argv
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None

B, N, M, NS, R = 8, 40000, 2048, 64, 0.2
dev = torch.device("cuda:0")
# --cloud U (default; SURVEY 8(d) cloud U(L)) | R (cloud R, room shell) | step (the point clouds of
# the batch bench.py's timed step runs on: votenet/data.py:make_batch, seed 100)
cloud = sys.argv[sys.argv.index("--cloud") + 1] if "--cloud" in sys.argv else "U"
xyz = bench.pair_cloud(cloud).to(dev)
flipped = xyz.transpose(1, 2).contiguous()
feat = torch.rand(B, 1, N, device=dev)
inds = ext.furthest_point_sampling(xyz, M)
new_xyz = ext.gather_points(flipped, inds).transpose(1, 2).contiguous()
torch.cuda.synchronize()


def api():
    idx = ext.ball_query(new_xyz, xyz, R, NS)
    ext.group_points(flipped, idx)
    ext.group_points(feat, idx)


def fused():
    ext.query_and_group(new_xyz, xyz, feat, R, NS, True)


# the SA layer's own sampling call and its centroids (pointnet2_modules.py:236-245)
LINDS, LISTS = ext.furthest_point_sampling_with_grid(xyz, M, R)
LNEW = ext.gather_points(flipped, LINDS).transpose(1, 2).contiguous()
assert torch.equal(LNEW, new_xyz)
if "--no-plan" not in sys.argv:
    LISTS.mark_centroids(LNEW, LINDS)


def layer():
    """as the set-abstraction layer runs it: cell lists left behind by the sampling kernel"""
    ext.query_and_group(LNEW, xyz, feat, R, NS, True, None, LISTS)


def build_only():
    ext.build_grid(xyz, R)


if plain:
    for _ in range(iters):
        if only in (None, "api"):
            api()
        if only in (None, "fused"):
            fused()
        if only in (None, "layer"):
            layer()
    if "--ablate" in sys.argv:  # counter passes of the timing ablations (kernel names differ)
        for a in ("1", "2"):
            os.environ["PN2_GRID_ABLATE"] = a
            for _ in range(iters):
                fused()
        os.environ.pop("PN2_GRID_ABLATE")
    torch.cuda.synchronize()
    print("pair_bench done (plain)")
else:
    res = {"shape": {"B": B, "N": N, "m": M, "nsample": NS, "radius": R}, "cloud": cloud,
           "algorithmic_bytes": bench.PAIR_BYTES}
    for name, fn in (("api", api), ("fused", fused), ("layer", layer), ("build_only", build_only)):
        if only is not None and name != only:
            continue
        us = bench.time_op(fn, iters=iters, warm=3)
        res[name + "_us"] = round(us, 2)
        res[name + "_frac_of_8TBs"] = round(bench.PAIR_BYTES / (us * 1e-6) / 8e12, 4)
    if "--sweep" in sys.argv:  # tuning switches / timing ablations of the cell-list kernels
        sweep = {}
        for t in ("1", "4"):
            os.environ["PN2_GRID_WPB"] = t
            sweep["layer_wpb%s_us" % t] = round(bench.time_op(layer, iters=iters, warm=2), 2)
        os.environ.pop("PN2_GRID_WPB")
        for a, what in (("1", "no_ordering_network"), ("2", "no_candidate_tests")):
            os.environ["PN2_GRID_ABLATE"] = a
            sweep["fused_%s_us" % what] = round(bench.time_op(fused, iters=iters, warm=2), 2)
        os.environ.pop("PN2_GRID_ABLATE")
        idx = ext.ball_query(new_xyz, xyz, R, NS)
        sweep["ball_query_us"] = round(bench.time_op(lambda: ext.ball_query(new_xyz, xyz, R, NS), iters=iters), 2)
        sweep["group_xyz_us"] = round(bench.time_op(lambda: ext.group_points(flipped, idx), iters=iters), 2)
        sweep["group_feat_us"] = round(bench.time_op(lambda: ext.group_points(feat, idx), iters=iters), 2)
        res["sweep"] = sweep
    print(json.dumps(res))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(res, f, indent=1)
