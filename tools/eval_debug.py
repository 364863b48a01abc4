"""Diagnostic: eval-mode (two-launch form) against train-mode activations unit by unit on the tiny model, with the gathering Focus stem
and with the im2col stem it replaced."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch
from ep24 import nn as enn, synth, engine as eengine
from ep24._lib import ptr
from ep24.options import PlanOptions, set_options
DEV = torch.device("cuda", 0)


def old_stem(self, focus):
    rows = self.new_act(112, self.IH // 2, self.IW // 2)
    rows.needs_grad = False
    self._f("stem_pack", ptr(self.images), rows.ptr(), 112, self.B, self.IH, self.IW)
    return self.unit(focus.conv, rows, stem=True)


def run(tag):
    torch.manual_seed(0)
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.125), enn.YOLOXHead(80, 0.125))
    m.head.initialize_biases(1e-2)
    m.to(DEV)
    set_options(m, PlanOptions(fold_bn_eval=False))
    B, S = 4, 128
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0
    eng = m.engine(B, S)
    images = synth.make_images(B, S, seed=3).to(DEV)
    m.train()
    out_train = m(images, train=True)[3].detach().clone()
    acts_train = {k: (v[1].buf.t.clone(), v[2].buf.t.clone()) for k, v in eng.unit_acts.items()}
    for mod, (x, z, out) in eng.unit_acts.items():
        Mrows = x.B * out.H * out.W
        mod.bn.running_var.mul_((Mrows - 1) / Mrows)
    m.eval()
    out_eval = m(images, train=False)
    ref = out_train.clone()
    ref[..., 26:] = torch.sigmoid(ref[..., 26:])
    print(tag, "final err %.4e" % float((out_eval - ref).abs().max() / ref.abs().max()), flush=True)
    names = {id(mod): n for n, mod in m.named_modules()}
    for k, v in eng.unit_acts.items():
        zt, ot = acts_train[k]
        ze, oe = v[1].buf.t, v[2].buf.t
        ez = float((ze.float() - zt.float()).abs().max() / (zt.float().abs().max() + 1e-20))
        eo = float((oe.float() - ot.float()).abs().max() / (ot.float().abs().max() + 1e-20))
        print("   %-44s z err %.3e  out err %.3e  M %d" % (names.get(id(k), str(type(k))), ez, eo, v[2].M), flush=True)


run("gather stem")
eengine.Engine.focus_stem = old_stem
run("im2col stem")
