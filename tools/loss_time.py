"""GPU box helper (round 5): the loss path's launches one by one - head outputs of a real YOLOX-l forward (B = 20, 640x640, 10 GTs),
every launch replayed 20 times from its own hipGraph, us per launch.  EP24_LIB selects the library (A/B against an older build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch
from ep24 import _lib, loss as eloss, nn as enn, train as etrain, synth
from ep24._lib import call, ptr, stream_ptr
DEV = torch.device("cuda", 0)
torch.manual_seed(0)
m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
m.head.initialize_biases(1e-2)
m.to(DEV)
ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=0.001, momentum=0.9, batch=20, size=640)
ts.eng.images.copy_(synth.make_images(20, 640, seed=1).to(DEV))
ts.labels.copy_(synth.make_labels(20, 10, size=640, seed=1000).to(DEV))
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
ws, eng = ts.ws, ts.eng
B, A, C = ws.B, ws.A, ws.C
ncols = 27 + C
in_box, in_ctr, match = ws.masks[0], ws.masks[1], ws.masks[2]
out, lab, xs, ys, st = eng.outputs, ts.labels, ts.xs, ts.ys, ts.st
steps = [
    ("assign_candidates", lambda s: call("assign_candidates", ptr(lab), ptr(xs), ptr(ys), ptr(st), ptr(ws.num_gt), ptr(in_box), ptr(in_ctr), B, A, s)),
    ("assign_cost", lambda s: call("assign_cost", ptr(out), ncols, ptr(lab), ptr(ws.num_gt), ptr(in_box), ptr(in_ctr), ptr(ws.pw), ptr(ws.cost), B, A, C, s)),
    ("memset match", lambda s: call("memset_zero", ptr(match), match.numel() * 8, s)),
    ("dynamic_k", lambda s: call("dynamic_k", ptr(ws.pw), ptr(ws.cost), ptr(ws.num_gt), ptr(in_box), ptr(in_ctr), ptr(match), ptr(ws.ks), B, A, s)),
    ("assign_resolve", lambda s: call("assign_resolve", ptr(match), ptr(ws.pw), ptr(ws.cost), ptr(ws.num_gt), ptr(ws.matched_gt), ptr(ws.matched_iou), B, A, s)),
    ("loss_terms", lambda s: call("loss_terms", ptr(out), ncols, ptr(lab), ptr(ws.matched_gt), ptr(ws.matched_iou), ptr(ws.partials), B, A, C, None, ptr(xs), ptr(ys), ptr(st), s)),
    ("loss_finalize", lambda s: call("loss_finalize", ptr(ws.partials), ws.nblocks, ptr(ws.num_gt), B, ptr(ts.state), ptr(ws.result), s)),
    ("loss_grad", lambda s: call("loss_grad", ptr(out), ncols, ptr(lab), ptr(ws.matched_gt), ptr(ws.matched_iou), ptr(ws.result), None, ptr(ws.dout), B, A, C, None,
                                 ptr(xs), ptr(ys), ptr(st), None, s)),
]
print("# library: %s" % _lib.LIB_PATH)
total = 0.0
keep_state = ts.state.clone()
for name, fn in steps:
    fn(stream_ptr())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn(stream_ptr())
    ts_ = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts_.append(e0.elapsed_time(e1) * 1e3 / 20)
    us = sorted(ts_)[2]
    total += us
    print("%-20s %8.1f us" % (name, us))
    ts.state.copy_(keep_state)
print("%-20s %8.1f us" % ("sum", total))
print("pairs with a lens / all (candidate, gt, ray) items: see csrc/assign.hip; num_fg %d" % int(ws.result[55]))
