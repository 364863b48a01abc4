"""GPU box helper: A/B of the halo-patch kernel against the generic tiled kernel on the 3x3 stride-1 layers of YOLOX-l
(B = 20), interleaved rounds in ONE process (median of 5), launches replayed from a hipGraph.
usage: conv_ab.py [fwd|dgrad|bnr ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24 import _lib  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
SHAPES = [(20, 40, 256, 256), (20, 80, 128, 128), (20, 20, 512, 512), (20, 80, 256, 256), (20, 160, 64, 64), (20, 80, 256, 512),
          (20, 40, 256, 512), (20, 20, 256, 256)]


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
    kinds = sys.argv[1:] or ["fwd", "dgrad"]
    fn = _lib.lib().fn
    print("%-8s %-22s %10s %10s %10s %10s" % ("kind", "B,H,Cin,Cout", "tiled us", "TF", "patch us", "TF"))
    for B, H, Cin, Cout in SHAPES:
        W = H
        x = torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16)
        w = (torch.randn(Cout, 9, Cin, device=DEV) * 0.05).to(torch.bfloat16)
        wd = (torch.randn(Cin, 9, Cout, device=DEV) * 0.05).to(torch.bfloat16)
        y = torch.zeros(B * H * W, Cout, device=DEV, dtype=torch.bfloat16)
        dy = torch.randn(B * H * W, Cout, device=DEV).to(torch.bfloat16)
        dx = torch.zeros(B * H * W, Cin, device=DEV, dtype=torch.bfloat16)
        z = torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16)
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        save = torch.ones(2, Cin, device=DEV)
        gam, bet = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
        sums = torch.zeros(2, Cin, dtype=torch.int64, device=DEV)
        fl = 2.0 * B * H * W * Cin * Cout * 9
        for kind in kinds:
            def run():
                if kind == "fwd":
                    call("conv_fwd_bf16", ptr(x), Cin, ptr(w), ptr(y), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, 3, 1, stream_ptr())
                elif kind == "dgrad":
                    call("conv_dgrad_bf16", ptr(dy), Cout, ptr(wd), ptr(dx), Cin, 0, B, H, W, Cin, Cout, 3, 1, stream_ptr())
                else:
                    call("conv_dgrad_bnr_bf16", ptr(dy), Cout, ptr(wd), ptr(dx), Cin, 0, B, H, W, Cin, Cout, 3, 1, ptr(z), Cin, ptr(save),
                         ptr(gam), ptr(bet), ptr(sums), ptr(sums, Cin), 1, stream_ptr())
            res = {}
            for rnd in range(2):
                for patch in (0, 1):
                    fn["ep24_conv_set_patch"](patch)
                    res.setdefault(patch, []).append(graph_time(run))
            fn["ep24_conv_set_patch"](1)
            t0, t1 = min(res[0]), min(res[1])
            print("%-8s %-22s %10.1f %10.1f %10.1f %10.1f" % (kind, "%d,%d,%d,%d" % (B, H, Cin, Cout), t0, fl / t0 / 1e6, t1, fl / t1 / 1e6), flush=True)


main()
