"""Diagnostic: per-parameter gradient of one full-size step with and without the fused BatchNorm-backward sums (PlanOptions.fuse_bn_reduce)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch
from ep24 import loss as eloss, nn as enn, train as etrain, synth
from ep24.options import PlanOptions, set_options
DEV = torch.device("cuda", 0)
TINY = "--tiny" in sys.argv
BATCH, SIZE = (4, 256) if TINY else (20, 640)


def run(plan, steps=1, lr=0.0, **kw):
    torch.manual_seed(0)
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.125), enn.YOLOXHead(80, 0.125)) if TINY else enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
    if "--nobias" not in sys.argv:
        m.head.initialize_biases(1e-2)
    m.to(DEV)
    set_options(m, PlanOptions.parse(plan))
    ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=lr, momentum=0.9, batch=BATCH, size=SIZE, **kw)
    if "--nobias" in sys.argv:
        ts.eng.images.copy_(synth.make_images(BATCH, SIZE, seed=3).to(DEV))
        ts.labels.copy_(synth.make_labels(BATCH, [3, 1, 6, 2], size=SIZE, seed=4).to(DEV))
    else:
        ts.eng.images.copy_(synth.make_images(BATCH, SIZE, seed=1).to(DEV))
        ts.labels.copy_(synth.make_labels(BATCH, 3 if TINY else 10, size=SIZE, seed=1000).to(DEV))
    for i in range(steps):
        loss = float(ts.step()[0])
        torch.cuda.synchronize()
        if TINY:
            continue
        mg = ts.ws.matched_gt.view(20, -1)
        d = ts.ws.dout.view(20, -1, 107)
        print("   step %d loss %.6f matched per level %d %d %d   |dout| level0 reg %.3e obj %.3e cls %.3e" % (
            i, loss, int((mg[:, :6400] >= 0).sum()), int((mg[:, 6400:8000] >= 0).sum()), int((mg[:, 8000:] >= 0).sum()),
            float(d[:, :6400, :26].abs().sum()), float(d[:, :6400, 26].abs().sum()), float(d[:, :6400, 27:].abs().sum())), flush=True)
    return loss, {n: p.grad.detach().float().clone() for n, p in m.named_parameters()}, ts


for tag, kw in (("graph", {}),) + (() if TINY else (("graph 3 steps lr 1e-3", dict(steps=3, lr=0.001)),)):
    la, ga, ts = run("fuse_bn_reduce=1", **kw)
    lb, gb, _ = run("fuse_bn_reduce=0", **kw)
    tot = sum(v.numel() for v in ga.values())
    print("zero fraction fused %.4f unfused %.4f" % (sum(float((v == 0).sum()) for v in ga.values()) / tot, sum(float((v == 0).sum()) for v in gb.values()) / tot))
    print(tag, "loss fused %.6f unfused %.6f" % (la, lb), "bnr launches", sum(1 for n, _ in ts.eng.bwd if "bnr" in n), flush=True)
    for n in ga:
        a, b = ga[n], gb[n]
        za, zb = float((a == 0).float().mean()), float((b == 0).float().mean())
        err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        if za > 0.5 or err > (1e-3 if TINY else 5e-2):
            print("  %-50s %-22s zero %.3f / %.3f  max rel diff %.3e" % (n, tuple(a.shape), za, zb, err), flush=True)

if "--repeat" in sys.argv:
    runs = [(plan, run(plan)) for plan in ("fuse_bn_reduce=1", "fuse_bn_reduce=1", "fuse_bn_reduce=0", "fuse_bn_reduce=0", "fuse_bn_reduce=1")]
    flat = [torch.cat([v.reshape(-1) for v in r[1][1].values()]) for r in runs]
    for i in range(len(runs)):
        for j in range(i + 1, len(runs)):
            d = float((flat[i] - flat[j]).abs().max() / flat[j].abs().max())
            print("run %d (%s) vs run %d (%s): equal %s  max diff / max %.3e" % (i, runs[i][0], j, runs[j][0], bool(torch.equal(flat[i], flat[j])), d), flush=True)

if "--sums" in sys.argv:
    from ep24.engine import STATS_REPLICAS as R
    out = {}
    for plan in ("fuse_bn_reduce=1", "fuse_bn_reduce=0"):
        _, _, ts = run(plan)
        eng = ts.eng
        names = {id(mod): n for n, mod in ts.model.named_modules()} if hasattr(ts, "model") else {}
        res, o = [], 0
        flat = eng.bnsums.view(-1).double().cpu()
        for spec in eng._sum_specs:
            C = spec // (2 * R)
            blk = flat[o:o + spec].view(R, 2, C).sum(0) / 2 ** 36
            res.append(blk)
            o += spec
        out[plan] = (res, [n for n, _ in eng.bwd])
    a, b = out["fuse_bn_reduce=1"][0], out["fuse_bn_reduce=0"][0]
    for i, (x, y) in enumerate(zip(a, b)):
        for w, nm in ((0, "sum du*zhat"), (1, "sum du")):
            d = (x[w] - y[w]).abs()
            print("unit %3d C %4d %-12s max |fused - unfused| %.3e   max |unfused| %.3e   rel %.2e" % (
                i, x.shape[1], nm, float(d.max()), float(y[w].abs().max()), float(d.max() / (y[w].abs().max() + 1e-30))), flush=True)
