"""GPU box helper: host time of every graph launch / event call of one captured training step (where does the host stall?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch
from ep24 import loss as eloss, nn as enn, train as etrain, synth
DEV = torch.device("cuda", 0)
torch.manual_seed(0)
m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
m.head.initialize_biases(1e-2)
m.to(DEV)
ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=0.001, momentum=0.9, batch=20, size=640)
ts.eng.images.copy_(synth.make_images(20, 640, seed=1).to(DEV))
ts.labels.copy_(synth.make_labels(20, 10, size=640, seed=1000).to(DEV))
for _ in range(5):
    ts.step()
torch.cuda.synchronize()
log = []
orig = torch.cuda.CUDAGraph.replay
names = {}
if isinstance(ts.g_fwd, tuple):
    for i, g in enumerate(ts.g_fwd):
        names[id(g)] = "fwd[%d]" % i
for i, (gm, gs, *_r) in enumerate(ts.g_bwd or []):
    if gm is not None: names[id(gm)] = "bwd main %d" % i
    if gs is not None: names[id(gs)] = "bwd side %d" % i
names[id(ts.g_upd)] = "update"


def timed(self):
    t0 = time.perf_counter()
    orig(self)
    log.append((names.get(id(self), "?"), t0, time.perf_counter()))


torch.cuda.CUDAGraph.replay = timed
for rep in range(2):
    del log[:]
    torch.cuda.synchronize()
    t00 = time.perf_counter()
    ts.step()
    t_host = time.perf_counter()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    print("step: host returned after %.2f ms, GPU done after %.2f ms" % ((t_host - t00) * 1e3, (t_end - t00) * 1e3))
    for n, a, b in log:
        print("   %-14s launch at +%6.2f ms, call took %6.3f ms" % (n, (a - t00) * 1e3, (b - a) * 1e3))
