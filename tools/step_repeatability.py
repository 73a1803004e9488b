"""How repeatable is the train step?  Runs the supervised step twice on two batches (the second
prefetched), several times eagerly and through the HIP graphs, and compares every run with the
first eager one: gradients after step 1 (same weights: a kernel check -- 5e-7 eager / eager,
7e-7 eager / graph), parameters after step 1 (which tensors moved differently), labels, loss and
gradients of step 2.  Finding of round 4 (profiles/r4_step_repeatability.txt): step 1 agrees to
round-off; the only parameters that leave step 1 differently are biases of BatchNorm layers whose
gradient is mathematically zero (shift invariance of the next BatchNorm) -- Adam turns their
round-off into +-lr -- and step 2's gradient then falls into one of a few populations.

    python tools/step_repeatability.py        # on the GPU box, from the repo root
"""
import importlib, sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
V = importlib.import_module("3dioumatch_amd.votenet")
data = importlib.import_module("3dioumatch_amd.votenet.data")
step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
dev = torch.device("cuda:0")
cfg = V.scannet_config()
B, N, K = 2, 20000, 64
import tests.test_train_step as T
B, N, K = T.B, T.N, T.K
batches = [{k: v.to(dev) for k, v in data.make_batch(B, N, cfg, seed=s, num_objects=5).items()} for s in (44, 45)]
def run(graphs):
    runner = V.SupervisedStep(cfg, dev, world_size=1, num_proposal=K, seed=3, graphs=graphs)
    if "--freeze" in sys.argv:  # as tests/test_train_step.py does: the known zero-gradient parameters
        step_mod.freeze_shift_invariant_parameters(runner.net)
    torch.manual_seed(9); torch.cuda.manual_seed_all(9)
    first, second = dict(batches[0]), dict(batches[1])
    runner.prefetch_geometry(first); runner.prefetch_geometry(second)
    l0, _ = runner(first)
    g_first = step_mod.flat_grads(runner.net).cpu().clone()
    p_first = [(n, q.detach().cpu().clone()) for n, q in runner.net.named_parameters()]
    l1, ep1 = runner(second)
    lab = (ep1['objectness_label'].float().cpu().clone(), ep1['objectness_mask'].float().cpu().clone(), ep1['object_assignment'].float().cpu().clone())
    names = [(n, p.grad.detach().cpu().clone() if p.grad is not None else None) for n, p in runner.net.named_parameters()]
    return float(l0), float(l1), g_first, step_mod.flat_grads(runner.net).cpu().clone(), step_mod.flat_params(runner.net).cpu().clone(), names, lab, p_first
rel = lambda a, b: float((a - b).norm() / b.norm())
ref = run(False)
def explain(x):
    worst = []
    for (n, a), (_, b) in zip(x[5], ref[5]):
        if a is None or b is None: continue
        worst.append((float((a - b).norm()), float(b.norm()), n, tuple(a.shape)))
    worst.sort(reverse=True)
    for w in worst[:8]: print("    d %.3e of %.3e  %s %s" % w)
for i in range(6):
    e = run(False); g = run(True)
    for tag, x in (("EAGER", e), ("GRAPH", g)):
        dl = [int((a != b).sum()) for a, b in zip(x[6], ref[6])]
        w = sorted(((float((a - b).abs().max()), float((a - b).norm() / (b.norm() + 1e-12)), n) for (n, a), (_, b) in zip(x[7], ref[7])), reverse=True)[:3]
        print("     params after step 1: " + "; ".join("%s max %.1e rel %.1e" % (n, m, r) for m, r, n in w))
        print("  %s run: g2 rel %.2e  label / mask / assignment entries differing from ref: %s  (positives %d vs %d)" % (
            tag, rel(x[3], ref[3]), dl, int(x[6][0].sum()), int(ref[6][0].sum())))
        if rel(x[3], ref[3]) > 1e-2 and "--explain" in sys.argv:
            explain(x)
    print("eager-eager: g1 %.2e g2 %.2e p %.2e | eager-graph: g1 %.2e g2 %.2e p %.2e" % (
        rel(e[2], ref[2]), rel(e[3], ref[3]), rel(e[4], ref[4]), rel(g[2], ref[2]), rel(g[3], ref[3]), rel(g[4], ref[4])))
