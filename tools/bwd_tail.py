"""GPU box helper (round 5): the last entries of the backward list with the graph segments they fall in - what the weight-gradient
lane still has to do when the main lane is through.  usage: bwd_tail.py [n_entries] [plan]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch
from ep24 import loss as eloss, nn as enn, train as etrain, synth
from ep24.options import PlanOptions, set_options
DEV = torch.device("cuda", 0)
n_show = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)
m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
m.head.initialize_biases(1e-2)
m.to(DEV)
set_options(m, PlanOptions.parse(sys.argv[2] if len(sys.argv) > 2 else ""))
ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=0.001, momentum=0.9, batch=20, size=640)
ts.eng.images.copy_(synth.make_images(20, 640, seed=1).to(DEV))
ts.labels.copy_(synth.make_labels(20, 10, size=640, seed=1000).to(DEV))
ts.step()
torch.cuda.synchronize()
eng = ts.eng
segs, _ = ts._segments()
seg_of = {}
for si, (lo, hi) in enumerate(segs):
    for i in range(lo, hi):
        seg_of[i] = si
print("# %d backward entries, %d segments: %s" % (len(eng.bwd), len(segs), segs[-8:]))
print("# early update cut:", ts._early_update_cut(segs), " chunks:", ts.update_chunks)
for i in range(max(0, len(eng.bwd) - n_show), len(eng.bwd)):
    e = eng.bwd[i]
    args = e[1] if len(e) > 1 else ()
    dims = [a for a in args if isinstance(a, int) and 0 < a < 10_000_000][:10]
    print("%4d seg %2d  %-34s %s" % (i, seg_of[i], e[0], dims))
