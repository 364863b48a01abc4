"""Debug helper (GPU box): first unit whose folded-BN eval output differs from the two-launch eval output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch
from ep24 import nn as enn, synth
from ep24.engine import Engine

torch.manual_seed(0)
W_ = float(os.environ.get("WIDTH", 0.25))
m = enn.YOLOX(enn.YOLOPAFPN(0.33, W_), enn.YOLOXHead(80, W_))
for mod in m.modules():
    if isinstance(mod, torch.nn.BatchNorm2d):
        torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
        torch.nn.init.uniform_(mod.bias, -0.2, 0.2)
        mod.running_mean.normal_(0, 0.1)
        mod.running_var.uniform_(0.5, 1.5)
m.to("cuda:0")
x = synth.make_images(2, 128, seed=3).to("cuda:0")
if os.environ.get("TRAIN_FIRST"):
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0
    m.train()
    m(x, train=True)
m.eval()
outs = {}
for fold in (0, 1):
    from ep24.options import PlanOptions, set_options
    set_options(m, PlanOptions(fold_bn_eval=bool(fold)))
    m._engines = {}
    eng = m.engine(2, 128)
    eng.forward_eval(x)
    torch.cuda.synchronize()
    names = {mod: n for n, mod in m.named_modules()}
    outs[fold] = [(names.get(k, str(k)), v[2].buf.t.clone(), v[2]) for k, v in eng.unit_acts.items()]
    final = eng.outputs.clone()
    outs[(fold, "out")] = final
for (n0, a, act0), (n1, b, act1) in zip(outs[0], outs[1]):
    va = a.view(act0.buf.rows, act0.buf.ld)[:, act0.c0:act0.c0 + act0.C].float()
    vb = b.view(act1.buf.rows, act1.buf.ld)[:, act1.c0:act1.c0 + act1.C].float()
    e = float((va - vb).abs().max() / (va.abs().max() + 1e-9))
    print("%-50s rel %.4f" % (n0, e))
print("final", float((outs[(0, "out")] - outs[(1, "out")]).abs().max() / outs[(0, "out")].abs().max()))
