"""Which decision of the semi-supervised step falls two ways at the same weights?

tests/test_ddp.py::test_graph_step_two_ranks_share_one_gpu[semi] sees, in about one run of three,
the step-0 gradient of a graph runner 1e-4 away from an eager runner's (same weights, same batch,
same seeds).  This script runs that test's configuration in ONE process: fresh runners, eager and
through the graphs alternately, one step each, and compares every run with the first eager one --
global gradient distance, the parameter tensors that differ most, and every tensor of the step's
end points that differs at all (the first one in the network's order names the decision).

Findings (round 5, profiles/r5_semi_step_branches.txt): alone on the GPU every run agrees to 4e-7.
With a second copy of this script running beside it: (1) jitter_center / jitter_size of the graph
runner differ in every entry in ~1 run of 9 -- the replay fills torch's per-generator (seed, offset)
pair once per graph, on two streams, and the later fill serves both graphs (since then the
runner draws the noise ahead of the graphs: expect none); (2) before
SemiSupervisedStep._stream_apart_from, the 7th graph runner's teacher stream WAS torch's default
capture stream (a pool of 32 handles, round-robin): shared BatchNorm tickets, gradient 1e6 apart.

    python tools/semi_step_branches.py [runs [rank]]    # on the GPU box, from the repo root
"""
import importlib, sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
V = importlib.import_module("3dioumatch_amd.votenet")
step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
unl = importlib.import_module("3dioumatch_amd.votenet.losses_unlabeled")
dev = torch.device(os.environ.get("SEMI_BRANCH_DEVICE", "cuda:0"))  # "cpu": host logic only, eager
cfg = V.scannet_config()
K = 32
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # the test's rank: batch seed 300 + rank, noise seed 100 + rank


def one(graphs):
    filt = unl.default_config_dict(cfg, unlabeled_batch_size=2)
    filt.update(obj_threshold=0.3, cls_threshold=0.03, iou_threshold=0.2)
    runner = V.SemiSupervisedStep(cfg, dev, world_size=1, num_proposal=K, graphs=graphs, config_dict=filt)
    step_mod.freeze_shift_invariant_parameters(runner.net)
    batch = V.make_semi_batch(1, 2, 4096, cfg, seed=300 + rank, num_objects=5, device=dev)
    torch.manual_seed(100 + rank)
    if dev.type == "cuda":
        torch.cuda.manual_seed_all(100 + rank)
    loss, ep = runner(batch)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    assert bool(runner.graphs) == graphs
    grads = [(n, p.grad.detach().float().cpu().clone()) for n, p in runner.net.named_parameters() if p.grad is not None]
    outs = {k: v.detach().float().cpu().clone() for k, v in ep.items() if torch.is_tensor(v)}
    return float(loss.detach()), step_mod.flat_grads(runner.net).cpu().clone(), grads, outs


ref = one(False)
print("reference (eager): loss %.6f, %d gradient tensors, %d end-point tensors" % (ref[0], len(ref[2]), len(ref[3])))
for i in range(runs):
    for tag, graphs in (("eager", False), ("graph", dev.type == "cuda")):
        x = one(graphs)
        g = float((x[1] - ref[1]).norm() / ref[1].norm())
        print("%s run %d: loss %+.2e apart, gradient %.2e apart" % (tag, i, x[0] - ref[0], g))
        if g < 1e-5:
            continue
        worst = sorted(((float((a - b).norm()), float(b.norm()), n) for (n, a), (_, b) in zip(x[2], ref[2])), reverse=True)
        for d, b, n in worst[:6]:
            print("    grad  d %.2e of %.2e  %s" % (d, b, n))
        diff = []
        for k, b in ref[3].items():
            a = x[3].get(k)
            if a is None or a.shape != b.shape:
                continue
            m = float((a - b).abs().max()) if a.numel() else 0.0
            if m > 0:
                diff.append((k, m, int((a != b).sum()), a.numel()))
        for k, m, c, n in diff:
            print("    out   %-40s max %.2e  %d of %d entries" % (k, m, c, n))
