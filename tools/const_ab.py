"""GPU box helper: the training-step rate with two module constants of ep24.engine changed in THIS process only (A/B of tunables that
are not plan options): usage const_ab.py STATS_REPLICAS=4 WGRAD_REDUCE_GROUP=8 ...   (no argument: the shipped values)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch
from ep24 import engine as eng_mod, loss as eloss, nn as enn, train as etrain, synth
for a in sys.argv[1:]:
    k, v = a.split("=")
    setattr(eng_mod, k, int(v))
DEV = torch.device("cuda", 0)
torch.manual_seed(0)
m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
m.head.initialize_biases(1e-2)
m.to(DEV)
ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=0.001, momentum=0.9, batch=20, size=640)
ts.eng.images.copy_(synth.make_images(20, 640, seed=1).to(DEV))
ts.labels.copy_(synth.make_labels(20, 10, size=640, seed=1000).to(DEV))
for _ in range(8):
    ts.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    ts.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
print("%-40s %.3f ms per step  %.1f images/s  loss %.5f" % (" ".join(sys.argv[1:]) or "shipped", dt * 1e3, 20 / dt, float(ts.ws.result[0])), flush=True)
