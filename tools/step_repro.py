"""Diagnostic: first-step loss of the full-size training step under different launch modes (lanes / graphs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch
from ep24 import loss as eloss, nn as enn, train as etrain, synth
DEV = torch.device("cuda", 0)


def run(lr, steps, plan="", **kw):
    from ep24.options import PlanOptions, set_options
    torch.manual_seed(0)
    m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
    m.head.initialize_biases(1e-2)
    m.to(DEV)
    set_options(m, PlanOptions.parse(plan))
    lf = eloss.Loss_Function(80)
    ts = etrain.TrainStep(m, lf, lr=lr, momentum=0.9, batch=20, size=640, **kw)
    ts.eng.images.copy_(synth.make_images(20, 640, seed=1).to(DEV))
    ts.labels.copy_(synth.make_labels(20, 10, size=640, seed=1000).to(DEV))
    losses = [float(ts.step()[0]) for _ in range(steps)]
    torch.cuda.synchronize()
    out = ts.eng.outputs.double()
    return losses, float(out.sum()), float(out.abs().sum())


for tag, kw, plan in (("graph 2-lane", {}, ""), ("graph 2-lane", {}, ""), ("eager", dict(use_graph=False), ""),
                      ("graph 1-lane fwd", {}, "parallel_forward=0"), ("graph no-graph-bwd", dict(graph_backward=False), ""),
                      ("graph 2-lane lr0", {}, "")):
    lr = 0.0 if tag.endswith("lr0") else 0.001
    print(tag, run(lr, 2, plan=plan, **kw), flush=True)
