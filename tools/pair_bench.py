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
(--plain: eager launches only, no graph capture -- what the counter passes want.)
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
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None

B, N, M, NS, R = 8, 40000, 2048, 64, 0.2
dev = torch.device("cuda:0")
xyz = torch.from_numpy(synth.cloud_uniform(B, N, synth.cube_side(N, R, NS), seed=1)).to(dev)
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


_, LISTS = ext.furthest_point_sampling_with_grid(xyz, M, R)  # the SA layer's own sampling call


def layer():
    """as the set-abstraction layer runs it: cell lists left behind by the sampling kernel"""
    ext.query_and_group(new_xyz, xyz, feat, R, NS, True, None, LISTS)


def build_only():
    ext.build_grid(xyz, R)


if plain:
    for _ in range(iters):
        api()
        fused()
        layer()
    if "--variants" in sys.argv:  # counter passes of the other query kernels (names differ)
        for variant, cpg in ((1, 0), (0, 4)):
            ext.grid_query_variant(variant, cpg)
            for _ in range(iters):
                layer()
        ext.grid_query_variant(2, 4)
    torch.cuda.synchronize()
    print("pair_bench done (plain)")
else:
    res = {"shape": {"B": B, "N": N, "m": M, "nsample": NS, "radius": R},
           "algorithmic_bytes": bench.PAIR_BYTES}
    for name, fn in (("api", api), ("fused", fused), ("layer", layer), ("build_only", build_only)):
        us = bench.time_op(fn, iters=iters, warm=3)
        res[name + "_us"] = round(us, 2)
        res[name + "_frac_of_8TBs"] = round(bench.PAIR_BYTES / (us * 1e-6) / 8e12, 4)
    if "--stages" in sys.argv:  # per-wave stage clocks of the grouped query kernel (s_memtime)
        stages = {}
        for cpg, fl in ((0, 0x42), (4, 0x40)):
            waves = B * ((M + 4 * cpg - 1) // (4 * cpg)) * 4 if cpg else 1024 * 8
            names = ("descriptors", "rows_wait+tests+list", "next_loads_issue", "ranking",
                     "slots+gather+stores_issue", "tail", "wave_total")
            if cpg == 0:
                names = ("tile_record+cell_offsets", "halo_copy+barrier", "ranges+tests+list", "ranking",
                         "slots+gather+stores_issue", "end_barrier", "wave_total")
            buf = torch.zeros(waves * 8, dtype=torch.int64, device=dev)
            ext.grid_query_profile(buf)
            ext.grid_query_variant(fl, cpg or 4)
            layer()
            torch.cuda.synchronize()
            ext.grid_query_variant(2, 4)
            ext.grid_query_profile(None)
            t = buf.view(waves, 8).cpu().double()
            if cpg == 0:  # waves 0 and 7 of every workgroup separately
                raw = buf.view(waves, 8)[:, 7].cpu()
                tiles, cen, gen = raw & 0xffff, (raw >> 16) & 0xffff, (raw >> 32) & 0xffff
                tot = t[:, 6]
                wg_tot = tot.view(-1, 8).max(dim=1).values
                wg_cen = cen.view(-1, 8).sum(dim=1)
                wg_gen = gen.view(-1, 8).sum(dim=1)
                wg_tiles = tiles.view(-1, 8)[:, 0]
                order_ = torch.argsort(wg_tot)
                pick = [int(order_[int(q * (len(order_) - 1))]) for q in (0.0, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0)]
                stages["tile_workgroups_by_time"] = [
                    {"quantile": q, "ticks": float(wg_tot[i]), "tiles": int(wg_tiles[i]),
                     "centroids": int(wg_cen[i]), "general_path": int(wg_gen[i])}
                    for q, i in zip((0.0, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0), pick)]
                stages["tile_totals"] = {"tiles": int(wg_tiles.sum()), "centroids": int(cen.sum()),
                                         "general_path": int(gen.sum())}
                for wv in (0, 7):
                    tw = t[wv::8]
                    stages["tile_wave%d" % wv] = {nm: round(float(tw[:, k].mean()), 1)
                                                  for k, nm in enumerate(names)}
            live = t[:, 6] > 0
            t = t[live]
            stages[("cpg%d" % cpg) if cpg else "tile"] = {
                "waves": int(live.sum()), "clock": "s_memtime ticks (shader cycles)",
                "mean_ticks": {nm: round(float(t[:, k].mean()), 1) for k, nm in enumerate(names)},
                "max_wave_total": float(t[:, 6].max())}
        res["stages"] = stages
        exp = {}
        keep = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)  # must outlive the graph capture
        for nm, fl in (("prof_kernel_baseline", 0x40), ("five_of_nine_row_loads", 0x50),
                       ("nine_loads_of_one_row", 0x140)):
            ext.grid_query_profile(keep)
            ext.grid_query_variant(fl, 4)
            exp[nm + "_us"] = round(bench.time_op(layer, iters=iters, warm=2), 2)
        ext.grid_query_variant(2, 4)
        ext.grid_query_profile(None)
        res["timing_experiments_wrong_results"] = exp
    if "--sweep" in sys.argv:  # kernel variants of the cell-list query (pn2_grid_query_variant)
        sweep = {}
        idx = ext.ball_query(new_xyz, xyz, R, NS)
        def layer_nofeat():
            ext.query_and_group(new_xyz, xyz, None, R, NS, True, None, LISTS)

        for name, (variant, cpg) in (("tile", (2, 4)), ("tile_nt", (0x22, 4)),
                                     ("round2_kernel", (1, 0)), ("grouped_cpg2", (0, 2)),
                                     ("grouped_cpg4", (0, 4)), ("grouped_cpg4_nt", (0x20, 4)),
                                     ("grouped_cpg2_nopf_8waves", (0x80, 2)),
                                     ("grouped_cpg4_nopf_8waves", (0x80, 4)),
                                     ("grouped_cpg2_nopf_8waves_nt", (0xa0, 2))):
            ext.grid_query_variant(variant, cpg)
            got_idx, got = ext.query_and_group(new_xyz, xyz, feat, R, NS, True, None, LISTS)
            sweep[name + "_idx_equal"] = bool(torch.equal(got_idx, idx))
            sweep[name + "_layer_us"] = round(bench.time_op(layer, iters=iters, warm=2), 2)
            sweep[name + "_layer_nofeat_us"] = round(bench.time_op(layer_nofeat, iters=iters, warm=2), 2)
            sweep[name + "_query_only_us"] = round(bench.time_op(
                lambda: ext.ball_query_prebuilt(new_xyz, xyz, R, NS, LISTS), iters=iters, warm=2), 2)
        ext.grid_query_variant(2, 4)
        sweep["ball_query_us"] = round(bench.time_op(lambda: ext.ball_query(new_xyz, xyz, R, NS), iters=iters), 2)
        sweep["group_xyz_us"] = round(bench.time_op(lambda: ext.group_points(flipped, idx), iters=iters), 2)
        sweep["group_feat_us"] = round(bench.time_op(lambda: ext.group_points(feat, idx), iters=iters), 2)
        res["sweep"] = sweep
    print(json.dumps(res))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(res, f, indent=1)
