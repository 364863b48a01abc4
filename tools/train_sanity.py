"""GPU box helper: overfit ONE synthetic batch with the captured step - the loss must fall steadily (training-dynamics sanity)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch
from ep24 import loss as eloss, nn as enn, synth, train as etrain

dev = torch.device("cuda", 0)
B, S = int(os.environ.get("B", 20)), 640
steps, lr = int(os.environ.get("STEPS", 300)), float(os.environ.get("LR", 0.002))
torch.manual_seed(0)
m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
m.head.initialize_biases(1e-2)
m.to(dev)
lf = eloss.Loss_Function(80)
ts = etrain.TrainStep(m, lf, lr=lr, momentum=0.9, batch=B, size=S)
ts.eng.images.copy_(synth.make_images(B, S, seed=1).to(dev))
ts.labels.copy_(synth.make_labels(B, 10, size=S, seed=1000).to(dev))
for i in range(steps):
    r = ts.step()
    if i % 20 == 0 or i == steps - 1:
        v = r.detach().float().cpu().tolist()
        print("step %4d  loss %.4f  iou(mean of 24) %.4f  obj %.4f  cls %.4f  num_fg %.0f" % (
            i, v[0], sum(v[1:25]) / 24, v[25], v[26], v[55]), flush=True)
