"""GPU box helper (round 5): where the forward pass's two lanes start and end in a NON-profiled run (rocprofv3's queue interception
resolves cross-queue waits itself, so its trace may show waits the plain runtime does not have).  Timed events at the lane
boundaries of ep24.train.TrainStep.step, median over replays; one line per PlanOptions.fwd_order given on the command line.
usage: fwd_lanes_probe.py [plan ...]   e.g.  fwd_lanes_probe.py "" fwd_order=1 fwd_order=2"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch
from ep24 import loss as eloss, nn as enn, train as etrain, synth
from ep24.options import PlanOptions, set_options
DEV = torch.device("cuda", 0)
plans = sys.argv[1:] or [""]
print("# ms from the step's first launch; median of 12 replays; YOLOX-l, B = 20, 640x640")
print("%-28s %8s %10s %9s %9s %10s %9s %8s" % ("plan", "fork", "side_begin", "main_end", "side_end", "loss_begin", "loss_end", "step"))
for plan in plans:
    torch.manual_seed(0)
    m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
    m.head.initialize_biases(1e-2)
    m.to(DEV)
    set_options(m, PlanOptions.parse(plan))
    ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=0.001, momentum=0.9, batch=20, size=640)
    ts.eng.images.copy_(synth.make_images(20, 640, seed=1).to(DEV))
    ts.labels.copy_(synth.make_labels(20, 10, size=640, seed=1000).to(DEV))
    for _ in range(6):
        ts.step()
    torch.cuda.synchronize()
    rec = {}
    for _ in range(12):
        ts.probe = {}
        ts.step()
        torch.cuda.synchronize()
        for k, e in ts.probe.items():
            if isinstance(e, list):
                for i, ei in enumerate(e):
                    rec.setdefault("%s[%d]" % (k, i), []).append(ts.probe["start"].elapsed_time(ei))
            elif k != "start":
                rec.setdefault(k, []).append(ts.probe["start"].elapsed_time(e))
    ts.probe = None
    med = {k: statistics.median(v) for k, v in rec.items()}
    if os.environ.get("EP24_PROBE_BWD"):
        nseg = len([k for k in med if k.startswith("bwd_main[")]) - 1
        print("#   backward, plan %r: segment i - main lane done at, side lane starts / ends its part at (ms); launches main / side" % (plan or "default"))
        for i in range(nseg):
            gm, gs = ts.g_bwd[i][0], ts.g_bwd[i][1]
            lo, hi = ts._segments()[0][i]
            nm, ns = [len(x) for x in ts.eng.lane_lists(lo, hi)]
            print("#   seg %2d  main %7.3f   side %7.3f .. %7.3f   (%3d / %3d launches)" % (
                i, med["bwd_main[%d]" % (i + 1)], med.get("bwd_side_begin[%d]" % i, float("nan")), med.get("bwd_side[%d]" % i, float("nan")), nm, ns))
        print("#   join at %s, backward ends %.3f, step %.3f" % (", ".join("%.3f" % v for k, v in sorted(med.items()) if k.startswith("bwd_join")), med["bwd_end[0]"], med["end"]))
    print("%-28s %8.3f %10.3f %9.3f %9.3f %10.3f %9.3f %8.3f" % (plan or "(default)", med["fork"], med["side_begin"], med["main_end"], med["side_end"],
                                                                  med["loss_begin"], med["loss_end"], med["end"]))
    del ts, m
    torch.cuda.empty_cache()
