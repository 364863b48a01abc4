"""GPU box helper: what ONE dependent launch costs inside a replayed hipGraph, whatever the kernel does - a chain of N ep24_memset_zero
launches over n bytes each, time per launch for a few n.  The intercept is the floor every one of the step's ~740 launches pays."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"


def chain_time(nbytes, n=400):
    buf = torch.empty(max(nbytes, 1024), dtype=torch.uint8, device=DEV)

    def run():
        call("memset_zero", ptr(buf), nbytes, stream_ptr())
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            run()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[2]


print("bytes per launch      us per launch (chain of 400 dependent ep24_memset_zero launches in one graph)")
for nb in (1024, 1 << 16, 1 << 20, 1 << 24, 1 << 26, 1 << 27):
    t = chain_time(nb)
    print("%12d %14.2f   %s" % (nb, t, "%.2f TB/s" % (nb / t / 1e6) if nb >= (1 << 24) else ""), flush=True)
