"""GPU box helper: two-stream vs single-stream backward of the -l model on identical inputs (gradient diff per segment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch
from ep24 import loss as eloss, nn as enn, synth, train as etrain

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
model.head.initialize_biases(1e-2)
model.to(dev)
lf = eloss.Loss_Function(80)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ts = etrain.TrainStep(model, lf, lr=0.0, momentum=0.9, batch=B, size=640, use_graph=False)
ts.eng.images.copy_(synth.make_images(B, 640, seed=1).to(dev))
ts.labels.copy_(synth.make_labels(B, 10, size=640, seed=1000).to(dev))
eng, home = ts.eng, ts.home
keep = [b.clone() for b in model.buffers()] + [ts.state.clone()]

def grads(side):
    eng.use_side = side
    with torch.no_grad():
        for b, k in zip(list(model.buffers()) + [ts.state], keep):
            b.copy_(k)
    ts._phase_forward()
    ts._phase_backward(0, len(eng.bwd))
    torch.cuda.synchronize()
    return home.gflat.clone(), float(ts.ws.result[0])

ref, l0 = grads(False)
ref2, l1 = grads(False)
print("loss", l0, l1, "single vs single max rel", float((ref - ref2).abs().max() / ref.abs().max()))
names = {}
for i in range(4):
    g, l = grads(True)
    d = (g - ref).abs()
    print("run", i, "loss", l, "finite", bool(torch.isfinite(g).all()), "max abs diff", float(d.max()), "ref max", float(ref.abs().max()))
    # worst segments
    worst = []
    pname = {id(p_): n for n, p_ in model.named_parameters()}
    for prm, seg in home.by_param.items():
        a = d[seg.off:seg.off + seg.numel] if hasattr(seg, "numel") else None
        if a is None:
            break
        r = float(a.max() / (ref[seg.off:seg.off + seg.numel].abs().max() + 1e-20))
        worst.append((r, pname.get(id(prm), '?'), seg.numel))
    worst.sort(reverse=True)
    print("   bad segments:", [w for w in worst if not (w[0] < 0.05)][:12])

# ---- hybrid (graph forward / eager two-stream backward / graph update with lr=0), many repetitions
eng.use_side = os.environ.get('SIDE', '1') == '1'
print('hybrid use_side', eng.use_side)
ts2 = etrain.TrainStep(model, lf, lr=0.0, momentum=0.9, batch=B, size=640, use_graph=True)
ts2.labels.copy_(ts.labels)
w0 = home.flat.clone()
img0 = eng.images.clone()
bad = 0
for i in range(int(os.environ.get("REPS", 6))):
    with torch.no_grad():
        for b, k in zip(list(model.buffers()) + [ts2.state], keep):
            b.copy_(k)
    mode = os.environ.get("MODE", "")
    if mode == "" or ts2.graphs is None:
        ts2.step()
    else:
        ts2.g_fwd.replay()
        if "nobwd" not in mode:
            ts2._phase_backward(0, len(eng.bwd))
        if "noupd" not in mode:
            ts2.g_upd.replay()
    torch.cuda.synchronize()
    g = ts2.home.gflat
    d = float((g - ref).abs().max())
    fin = bool(torch.isfinite(g).all())
    print("hybrid rep", i, "loss", float(ts2.ws.result[0]), "finite", fin, "max abs diff", d,
          "| weights changed", int((home.flat != w0).sum()), "nonfinite w", int((~torch.isfinite(home.flat)).sum()),
          "images changed", int((eng.images != img0).sum()), "labels changed", int((ts2.labels != ts.labels).sum()),
          "state diff", float((ts2.state - keep[-1]).abs().max()), "mom max", float(home.mflat.abs().max()))
    if i == 1 and os.environ.get("MODE"):
        snap = dict(out=eng.outputs.clone(), stats=eng.stats.clone(), ng=ts2.ws.num_gt.clone(), mg=ts2.ws.matched_gt.clone(),
                    part=ts2.ws.partials.clone(), res=ts2.ws.result.clone(), wf=home.wf.clone(), st=ts2.state.clone())
        with torch.no_grad():
            for b, k in zip(list(model.buffers()) + [ts2.state], keep):
                b.copy_(k)
        ts2._phase_forward(); torch.cuda.synchronize()
        cur = dict(out=eng.outputs, stats=eng.stats, ng=ts2.ws.num_gt, mg=ts2.ws.matched_gt, part=ts2.ws.partials,
                   res=ts2.ws.result, wf=home.wf, st=ts2.state)
        print("   stats graph", snap["stats"][:4].tolist(), "eager", eng.stats[:4].tolist(), "ratio", (snap["stats"][:8].double() / eng.stats[:8].double()).tolist())
        for k_ in snap:
            a_, b_ = snap[k_].double(), cur[k_].double()
            print("   graph-vs-eager", k_, "max abs diff", float((a_ - b_).abs().max()), "n diff", int((a_ != b_).sum()), "of", a_.numel())
    if i == 2:
        # eager forward with whatever the weights are now
        with torch.no_grad():
            for b, k in zip(list(model.buffers()) + [ts2.state], keep):
                b.copy_(k)
        ts._phase_forward(); torch.cuda.synchronize()
        print("   eager forward loss now", float(ts.ws.result[0]))
