"""Debug helper (GPU box): per-layer comparison of the HIP plan against the bf16-emulating oracle."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
from ep24 import nn as enn, synth
from oracle import model as om

def act_nchw(a):
    return a.buf.t.view(a.buf.rows, a.buf.ld)[:, a.c0:a.c0 + a.C].reshape(a.B, a.H, a.W, a.C).permute(0, 3, 1, 2).float().cpu()

def gact_nchw(a):
    r = a._groot()
    return r.buf.grad().view(r.buf.rows, r.buf.ld)[:, r.c0:r.c0 + r.C].reshape(a.B, a.H, a.W, a.C).permute(0, 3, 1, 2).float().cpu()

def main():
    depth, width, B, S = 0.33, 0.25, 4, 256
    torch.manual_seed(3)
    ref = om.Net(depth, width)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5); torch.nn.init.uniform_(mod.bias, -0.2, 0.2)
    m = enn.YOLOX(enn.YOLOPAFPN(depth, width), enn.YOLOXHead(80, width))
    m.load_state_dict(ref.state_dict(), strict=True)
    m.to("cuda:0")
    x = synth.make_images(B, S, seed=9)
    rec = {}
    names = {mod: n for n, mod in ref.named_modules()}
    def hook(mod, inp, out):
        rec[names[mod]] = (inp[0].detach(), out.detach())
    for mod in ref.modules():
        if isinstance(mod, om.Unit):
            mod.register_forward_hook(hook)
    om.EMULATE_BF16 = True
    ref.train()
    o_ref = ref(x, train=True)[3]
    out = m(x.to("cuda:0"), train=True)[3]
    eng = m.engine(B, S)
    pn = {mod: n for n, mod in m.named_modules()}
    for mod, (xin, z, y) in eng.unit_acts.items():
        n = pn[mod]
        rin, rout = rec[n]
        got = act_nchw(y)
        e_out = float((got - rout).abs().max() / (rout.abs().max() + 1e-9))
        if xin.C == rin.shape[1]:
            e_in = float((act_nchw(xin) - rin).abs().max() / (rin.abs().max() + 1e-9))
        else:
            e_in = -1
        print("%-46s in %.4f out %.4f  shape %s" % (n, e_in, e_out, tuple(rout.shape)))
    print("final", float((out.cpu() - o_ref).abs().max()), float(o_ref.abs().max()))

if __name__ == "__main__":
    main()
