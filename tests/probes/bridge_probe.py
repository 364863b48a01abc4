"""GPU box helper: the bf16 bridge table (tests/bridge_stages.py) at a given batch.  usage: bridge_probe.py [BATCH ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bridge_stages as bs  # noqa: E402
from ep24 import synth  # noqa: E402

for B in [int(v) for v in sys.argv[1:]] or [20]:
    ref, m = bs.build_pair()
    t0 = time.time()
    rows, _, _ = bs.bridge_table(ref, m, synth.make_images(B, 640, seed=9))
    print("B = %d, 640 x 640 (%.0f s)" % (B, time.time() - t0))
    print("%-16s %5s | %12s %12s %12s | %12s %12s | %12s %12s" % ("stage", "units", "tf rms", "rms bound", "tf max/range", "chained rms", "ch max/range", "propagated", "ch bound"))
    for st, n, a, b, c, d, pr in rows:
        print("%-16s %5d | %12.3e %12.3e %12.3e | %12.3e %12.3e | %12s %12.3e" % (st, n, a, bs.rms_bound(st), b, c, d, "-" if pr is None else "%.3e" % pr,
                                                                            bs.rms_bound(st) + bs.PROP_SLACK * (pr or 0.0)), flush=True)
