import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
importlib.import_module("3dioumatch_amd")
V = importlib.import_module("3dioumatch_amd.votenet")
data = importlib.import_module("3dioumatch_amd.votenet.data")
dev = torch.device("cuda:0")
cfg = V.scannet_config()
batch = data.make_batch(8, 40000, cfg, seed=100, device=dev)
runner = V.SupervisedStep(cfg, dev, world_size=1, num_proposal=256, lr=1e-3)
names = [n for n, _ in runner.net.named_parameters()]
sizes = [p.numel() for p in runner.net.parameters()]
p0 = runner.flat_params.data.clone()
b0 = [b.clone() for b in runner.net.buffers()]
def seg_bad(flat):
    out, off = [], 0
    for n, s in zip(names, sizes):
        if not torch.isfinite(flat[off:off + s]).all():
            out.append(n)
        off += s
    return out
for trial in range(int(os.environ.get("TRIALS", "10"))):
    torch.cuda.synchronize()
    runner.flat_params.data.copy_(p0)
    for b, s in zip(runner.net.buffers(), b0):
        b.copy_(s)
    for v in runner.optimizer.state.get(runner.flat_params, {}).values():
        if torch.is_tensor(v):
            v.zero_()
    views = [dict(batch), dict(batch)]
    if os.environ.get("DBG_NOPRE", "0") != "1":
        runner.prefetch_geometry(views[0])
    for i in range(14):
        if os.environ.get("DBG_NOPRE", "0") != "1":
            runner.prefetch_geometry(views[(i + 1) % 2])
        if os.environ.get("DBG_SYNC_PRE", "0") == "1":
            torch.cuda.synchronize()
        loss, ep = runner(views[i % 2])
        torch.cuda.synchronize()
        gb, pb = seg_bad(runner.flat_grad), seg_bad(runner.flat_params.data)
        if gb or pb or not torch.isfinite(loss):
            print("trial", trial, "step", i, "loss", float(loss))
            print("   bad grads :", len(gb), "of", len(names), " good:", [n for n in names if n not in gb][:40])
            print("   bad params:", len(pb))
            bad_ep = [k for k, v in ep.items() if torch.is_tensor(v) and v.is_floating_point() and not torch.isfinite(v).all()]
            print("   bad end_points:", bad_ep[:20], flush=True)
            break
print("done")
