"""Diagnostic: replay the full-size captured training step many times on the same batch with frozen weights (lr = 0, the
loss's running state restored before every step) and compare an exact integer checksum of EVERY engine buffer
(activations, gradients, BN sums, outputs, loss workspace, flat gradient) with the first step.  Any difference is a
run-to-run nondeterminism; the first differing buffer in creation (= forward) order names the producing layer.

    python tools/step_stress.py [--steps 400] [--width 1.0 --depth 1.0 --batch 20 --size 640] [--eager]"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch
from ep24 import engine as eengine, loss as eloss, nn as enn, train as etrain, synth

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--width", type=float, default=1.0)
ap.add_argument("--depth", type=float, default=1.0)
ap.add_argument("--batch", type=int, default=20)
ap.add_argument("--size", type=int, default=640)
ap.add_argument("--eager", action="store_true")
ap.add_argument("--backbone", default="darknet", choices=["darknet", "resnet", "densenet", "vgg"])
ap.add_argument("--kernel-opts", type=int, default=0, help="PlanOptions.conv_kernel_opts: bit0 tiled kernel instead of the halo-patch kernel, bit1 narrow epilogue")
a = ap.parse_args()
DEV = torch.device("cuda", 0)

BUFS = []
_init = eengine.Buf.__init__


def _rec(self, *args, **kw):
    _init(self, *args, **kw)
    BUFS.append(self)


eengine.Buf.__init__ = _rec

from ep24.options import PlanOptions, set_options
torch.manual_seed(0)
m = enn.YOLOX(enn.YOLOPAFPN(a.depth, a.width, backbone_type=a.backbone), enn.YOLOXHead(80, a.width))
m.head.initialize_biases(1e-2)
m.to(DEV)
set_options(m, PlanOptions(conv_kernel_opts=a.kernel_opts))
lf = eloss.Loss_Function(80)
ts = etrain.TrainStep(m, lf, lr=0.0, momentum=0.9, batch=a.batch, size=a.size, use_graph=not a.eager)
ts.eng.images.copy_(synth.make_images(a.batch, a.size, seed=1).to(DEV))
ts.labels.copy_(synth.make_labels(a.batch, 10, size=a.size, seed=1000).to(DEV))
eng = ts.eng
if getattr(eng, "drop_keep", None) is not None:
    eng.fixed_dropout = True                                # DenseNet: the same Dropout2d draws in every step
BUF_ID = {id(b): i for i, b in enumerate(BUFS)}
UNIT_OF = {}                                                # buffer index -> "layer name (role)"
_names = {id(mod): n for n, mod in m.named_modules()}
for key, (xin, z, y) in eng.unit_acts.items():
    nm = _names.get(id(key), type(key).__name__)
    for role, act in (("x", xin), ("z", z), ("y", y)):
        if act is not None:
            UNIT_OF.setdefault(BUF_ID.get(id(act.buf), -1), "%s.%s" % (nm, role))


def tensors():
    out = []
    for i, b in enumerate(BUFS):
        out.append(("buf%03d[%dx%d]" % (i, b.rows, b.ld), b.t))
        if b.g is not None:
            out.append(("buf%03d.grad" % i, b.g))
    out += [("outputs", eng.outputs), ("stats", eng.stats), ("bnsums", eng.bnsums), ("dzbuf", eng.dzbuf), ("slab", eng.slab),
            ("gflat", ts.home.gflat), ("flat", ts.home.flat), ("result", ts.ws.result)]
    for k, v in vars(ts.ws).items():
        if torch.is_tensor(v) and v.is_cuda:
            if k == "masks":
                out += [("ws.in_box", v[0]), ("ws.in_ctr", v[1]), ("ws.match", v[2])]
            else:
                out.append(("ws." + k, v))
    return out


def checksum(t):
    t = t.reshape(-1)
    if t.dtype == torch.int64:
        return t.sum()
    nb = t.numel() * t.element_size()
    if nb % 4 == 0 and t.data_ptr() % 4 == 0:
        return t.view(torch.int32).sum(dtype=torch.int64)
    return t.view(torch.uint8).sum(dtype=torch.int64)


state0 = ts.state.clone()
ref, names, prev = None, None, None
bad = 0
for step in range(a.steps):
    ts.state.copy_(state0)
    ts.step()
    tl = tensors()
    cs = torch.stack([checksum(t) for _, t in tl])
    if ref is None:
        ref, names, prev = cs.clone(), [n for n, _ in tl], cs.clone()
        full = {n: t.clone() for n, t in tl if n.startswith("ws.")}
        print("tracking %d tensors, %.2f GB" % (len(tl), sum(t.numel() * t.element_size() for _, t in tl) / 1e9), flush=True)
        continue
    diff = (cs != ref).nonzero().flatten().tolist()
    if diff and bad == 0:
        same = [names[i] for i in range(len(names)) if i not in set(diff)]
        print("unchanged tensors: %s" % ", ".join(same), flush=True)
        prev_diff = (cs != prev).nonzero().flatten().tolist() if prev is not None else None
    elif diff:
        prev_diff = (cs != prev).nonzero().flatten().tolist()
        print("step %d vs the step before it: %d tensors differ" % (step, len(prev_diff)), flush=True)
    prev = cs.clone()
    if diff:
        bad += 1
        print("step %d: %d tensors differ; first: %s" % (step, len(diff), ", ".join(names[i] for i in diff[:12])), flush=True)
        for i in diff[:6]:
            if names[i].startswith("buf") and names[i][3:6].isdigit():
                print("   %s = %s" % (names[i], UNIT_OF.get(int(names[i][3:6]), "?")), flush=True)
        for i in diff[:4]:
            n = names[i]
            if n in full:
                cur = dict(tl)[n]
                idx = (cur != full[n]).nonzero()
                print("   %s: %d elements differ; first %s: was %s now %s" % (
                    n, idx.shape[0], idx[:3].tolist(), [full[n][tuple(j)].item() for j in idx[:3]], [cur[tuple(j)].item() for j in idx[:3]]), flush=True)
        if bad >= 8:
            break
    if step % 50 == 0:
        print("step %d ok so far (%d bad)" % (step, bad), flush=True)
if os.environ.get("EP24_LIB", "").endswith("stamps.so"):
    import ctypes
    h = (ctypes.c_ulonglong * 16)()
    torch.cuda.synchronize()
    dll = ctypes.CDLL(os.environ["EP24_LIB"])
    dll.ep24_debug_read_cand(h, 16)
    v = list(h)
    print("deg-level deviations by pass: product-shape LDS %d | LDS read + s_nop %d | registers only (preloaded) %d | global %d | no majority %d; lanes outside 48..63: %d; threads %d"
          % (v[0], v[1], v[2], v[3], v[7], v[9], v[8]))
print("done: %d steps, %d differed from step 0; loss %r" % (step + 1, bad, float(ts.ws.result[0])))
