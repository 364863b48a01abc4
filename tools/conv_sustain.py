"""GPU box helper: one conv layer launched back to back for seconds (a graph of 200 launches replayed), i.e. the time per
launch at the clock the chip HOLDS under that load, not the burst clock of a 1 ms measurement.
usage: [EP24_LIB=...] conv_sustain.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
for B, H, Cin, Cout, k, s in [(20, 40, 256, 256, 3, 1), (20, 80, 256, 256, 3, 1), (20, 20, 512, 512, 3, 1), (20, 80, 256, 512, 3, 2), (20, 40, 512, 512, 1, 1)]:
    W = H
    Ho = H // s
    x = torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(Cout, k * k, Cin, device=DEV) * 0.05).to(torch.bfloat16)
    y = torch.zeros(B * Ho * Ho, Cout, device=DEV, dtype=torch.bfloat16)
    stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)

    def run():
        call("conv_fwd_bf16", ptr(x), Cin, ptr(w), ptr(y), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, k, s, stream_ptr())

    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(200):
            run()
    g.replay()
    torch.cuda.synchronize()
    n = 0
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < SECONDS:
        g.replay()
        n += 200
        if n % 2000 == 0:
            torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 2.0 * B * Ho * Ho * Cin * Cout * k * k
    print("%-22s %8.1f us  %7.1f TF  (%d launches)" % ("%d,%d,%d,%d,%d,%d" % (B, H, Cin, Cout, k, s), us, fl / us / 1e6, n), flush=True)
