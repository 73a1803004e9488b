import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
importlib.import_module("3dioumatch_amd")
V = importlib.import_module("3dioumatch_amd.votenet")
data = importlib.import_module("3dioumatch_amd.votenet.data")
dev = torch.device("cuda:0")
cfg = V.scannet_config()
batch = data.make_batch(8, 40000, cfg, seed=100, device=dev)
trials = int(os.environ.get("TRIALS", "6"))
runner = V.SupervisedStep(cfg, dev, world_size=1, num_proposal=256, lr=1e-3)
p0 = runner.flat_params.data.clone()
b0 = [b.clone() for b in runner.net.buffers()]
KEYS = ["vote_loss", "objectness_loss", "box_loss", "sem_cls_loss", "iou_loss", "jitter_iou_loss"]
nbad = 0
SYNC_PRE = os.environ.get("DBG_SYNC_PRE", "0") == "1"
with torch.no_grad():
    ref = {k: v.clone() for k, v in runner.net.compute_geometry(batch).items() if torch.is_tensor(v)}
torch.cuda.synchronize()
GK = sorted(ref)
GEO = os.environ.get('DBG_GEO', '1') == '1'
FENCE = [int(x) for x in os.environ.get("DBG_FENCE", "3").split(",") if x]
NOPRE = os.environ.get("DBG_NOPRE", "0") == "1"
for trial in range(trials):
    torch.cuda.synchronize()
    runner.flat_params.data.copy_(p0)
    for b, s in zip(runner.net.buffers(), b0):
        b.copy_(s)
    for v in runner.optimizer.state.get(runner.flat_params, {}).values():
        if torch.is_tensor(v):
            v.zero_()
    views = [dict(batch), dict(batch)]
    if not NOPRE:
        runner.prefetch_geometry(views[0])
    rec = []
    geo_ok = []
    for i in range(14):
        if i in FENCE:
            torch.cuda.synchronize()
        if NOPRE:
            loss, ep = runner(views[i % 2])
            rec.append(torch.stack([loss.detach()] + [ep[k].detach() for k in KEYS]))
            continue
        runner.prefetch_geometry(views[(i + 1) % 2])
        if SYNC_PRE:
            torch.cuda.synchronize()
        loss, ep = runner(views[i % 2])
        rec.append(torch.stack([loss.detach()] + [ep[k].detach() for k in KEYS]))
        if runner.graphs and GEO:
            geo_ok.append(torch.stack([(runner._cur_geometry[k] == ref[k]).all() for k in GK]))
    torch.cuda.synchronize()
    rec = torch.stack(rec).cpu()
    if not torch.isfinite(rec).all():
        nbad += 1
        i = int((~torch.isfinite(rec).all(dim=1)).nonzero()[0])
        print("trial", trial, "first non-finite at step", i)
        print("   this:", rec[i].tolist(), flush=True)
        if geo_ok:
            g = torch.stack(geo_ok).cpu()
            for st in range(g.shape[0]):
                if not g[st].all():
                    print("   step", st, "geometry mismatch:", [GK[j] for j in range(len(GK)) if not g[st, j]])
    elif geo_ok and not torch.stack(geo_ok).all():
        print("trial", trial, "finite but geometry mismatch")
print("graphs", runner.graphs, "trials", trials, "bad", nbad)
