import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/exploration-of-potential_amd") else os.getcwd()
ROOT = os.environ.get("GRAFT_REPO_ROOT", ROOT)
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch
from ep24 import loss as eloss, nn as enn, train as etrain, synth
DEV = torch.device("cuda", 0)
torch.manual_seed(0)
m = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.25), enn.YOLOXHead(80, 0.25))
m.head.initialize_biases(1e-2)
m.to(DEV)
B, S, G = 8, 320, int(sys.argv[2])
ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=0.0, momentum=0.9, batch=B, size=S)
ts.eng.images.copy_(synth.make_images(B, S, seed=1).to(DEV))
ts.labels.copy_(synth.make_labels(B, G, size=S, seed=1000).to(DEV))
ts.step()
torch.cuda.synchronize()
ws = ts.ws
cand = ((ws.masks[0] | ws.masks[1]) != 0)
torch.save({"pw": ws.pw.cpu(), "cost": ws.cost.cpu(), "cand": cand.cpu(), "num_gt": ws.num_gt.cpu(), "loss": ws.result.cpu(), "mg": ws.matched_gt.cpu()}, sys.argv[1])
print("saved", sys.argv[1], float(ws.result[0]), int(cand.sum()))
