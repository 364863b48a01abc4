"""GPU box helper: time the conv entry points on the YOLOX-l (B = 20) layer shapes, hipGraph-replayed, median of 5.
usage: conv_time.py [all|3x3|1x1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
S3 = [(20, 40, 256, 256, 3, 1), (20, 80, 128, 128, 3, 1), (20, 20, 512, 512, 3, 1), (20, 80, 256, 256, 3, 1), (20, 160, 64, 64, 3, 1),
      (20, 320, 64, 128, 3, 2), (20, 160, 128, 256, 3, 2), (20, 80, 256, 512, 3, 2), (20, 40, 512, 1024, 3, 2)]
S1 = [(20, 40, 256, 256, 1, 1), (20, 80, 128, 128, 1, 1), (20, 80, 256, 256, 1, 1), (20, 40, 512, 512, 1, 1), (20, 160, 64, 64, 1, 1),
      (20, 20, 1024, 1024, 1, 1), (20, 20, 512, 512, 1, 1), (20, 40, 512, 256, 1, 1), (20, 20, 2048, 1024, 1, 1), (20, 20, 1024, 512, 1, 1),
      (20, 160, 128, 128, 1, 1), (20, 320, 112, 64, 1, 1), (20, 80, 512, 256, 1, 1), (20, 40, 1024, 512, 1, 1)]


def graph_time(run, iters=20):
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
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
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(ts)[2]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    shapes = (S3 if which in ("all", "3x3") else []) + (S1 if which in ("all", "1x1") else [])
    print("%-28s %9s %8s | %9s %8s | %9s %8s" % ("B,H,Cin,Cout,k,s", "fwd us", "TF", "dgrad us", "TF", "wgrad us", "TF"))
    tot = [0.0, 0.0, 0.0]
    for B, H, Cin, Cout, k, s in shapes:
        W = H
        OH = (H - 1) // s + 1
        x = torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16)
        w = (torch.randn(Cout, k * k, Cin, device=DEV) * 0.05).to(torch.bfloat16)
        wd = (torch.randn(Cin, k * k, Cout, device=DEV) * 0.05).to(torch.bfloat16)
        y = torch.zeros(B * OH * OH, Cout, device=DEV, dtype=torch.bfloat16)
        dy = torch.randn(B * OH * OH, Cout, device=DEV).to(torch.bfloat16)
        dx = torch.zeros(B * H * W, Cin, device=DEV, dtype=torch.bfloat16)
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        from ep24 import _lib
        splits = _lib.lib().fn["ep24_conv_wgrad_splits"](B, H, W, Cin, Cout, k, s)
        slab = torch.zeros(splits * Cout * k * k * Cin, device=DEV)
        fl = 2.0 * B * OH * OH * Cin * Cout * k * k
        f = lambda: call("conv_fwd_bf16", ptr(x), Cin, ptr(w), ptr(y), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, k, s, stream_ptr())
        d = lambda: call("conv_dgrad_bf16", ptr(dy), Cout, ptr(wd), ptr(dx), Cin, 0, B, H, W, Cin, Cout, k, s, stream_ptr())
        g = lambda: call("conv_wgrad_slab_bf16", ptr(x), Cin, ptr(dy), Cout, ptr(slab), slab.numel(), k * k * Cin, Cout, Cin, B, H, W, Cin, Cout, k, s, stream_ptr())
        t = [graph_time(f), graph_time(d), graph_time(g)]
        for i in range(3):
            tot[i] += t[i]
        print("%-28s %9.1f %8.1f | %9.1f %8.1f | %9.1f %8.1f" % ("%d,%d,%d,%d,%d,%d" % (B, H, Cin, Cout, k, s), t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6,
                                                              t[2], fl / t[2] / 1e6), flush=True)
    print("sum us: fwd %.1f dgrad %.1f wgrad %.1f" % tuple(tot))


main()
