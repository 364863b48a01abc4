"""GPU box helper: A/B of the ring without a patch (conv_ring_generic_kernel, kernel_opts bit 4) against the tiled kernel (the default)
on the layers of YOLOX-l (B = 20) that fit it: 1x1 layers with K > 128 that do not stream, stride-2 3x3
layers.  Interleaved rounds in ONE process, launches replayed from a hipGraph; operands rotate over several buffer sets so that a
launch does not find its inputs in L2 from the launch before.
usage: gring_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24 import _lib  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
SHAPES = [(20, 40, 256, 256, 1, 1), (20, 40, 512, 512, 1, 1), (20, 40, 512, 256, 1, 1), (20, 40, 1024, 512, 1, 1), (20, 20, 1024, 1024, 1, 1),
          (20, 20, 2048, 1024, 1, 1), (20, 80, 512, 256, 1, 1), (20, 320, 64, 128, 3, 2), (20, 160, 128, 256, 3, 2), (20, 80, 256, 512, 3, 2),
          (20, 40, 512, 1024, 3, 2), (20, 80, 256, 256, 3, 2)]
SETS = 6


def graph_time(run, iters=18):
    run(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            run(i % SETS)
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
    fn = _lib.lib().fn
    print("%-6s %-26s %8s %10s %10s   (us; TFLOP/s)" % ("kind", "B,H,Cin,Cout,k,s", "kernel", "tiled", "ring"))
    for B, H, Cin, Cout, k, s in SHAPES:
        W = H
        OH = (H - 1) // s + 1
        T = k * k
        xs = [torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16) for _ in range(SETS)]
        w = (torch.randn(Cout, T, Cin, device=DEV) * 0.05).to(torch.bfloat16)
        wd = (torch.randn(Cin, T, Cout, device=DEV) * 0.05).to(torch.bfloat16)
        ys = [torch.zeros(B * OH * OH, Cout, device=DEV, dtype=torch.bfloat16) for _ in range(SETS)]
        dys = [torch.randn(B * OH * OH, Cout, device=DEV).to(torch.bfloat16) for _ in range(SETS)]
        dxs = [torch.zeros(B * H * W, Cin, device=DEV, dtype=torch.bfloat16) for _ in range(SETS)]
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        fl = 2.0 * B * OH * OH * Cin * Cout * T
        kid = fn["ep24_conv_kernel_for_ex"](0, B, H, W, Cin, Cout, k, s, 0, 0, 16)
        for kind in (("fwd", "dgrad") if s == 1 else ("fwd",)):
            ko = [0]

            def run(i):
                if kind == "fwd":
                    call("conv_fwd_bf16_ex", ptr(xs[i]), Cin, ptr(w), ptr(ys[i]), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, k, s, ko[0], stream_ptr())
                else:
                    call("conv_dgrad_bf16_ex", ptr(dys[i]), Cout, ptr(wd), ptr(dxs[i]), Cin, 0, B, H, W, Cin, Cout, k, s, ko[0], stream_ptr())
            res = {}
            for rnd in range(3):
                for mode in (0, 16):
                    ko[0] = mode
                    res.setdefault(mode, []).append(graph_time(run))
            print("%-6s %-26s %8d %10.1f %10.1f   %5.0f %5.0f" % (kind, "%d,%d,%d,%d,%d,%d" % (B, H, Cin, Cout, k, s), kid, min(res[0]), min(res[16]),
                                                               fl / min(res[0]) / 1e6, fl / min(res[16]) / 1e6), flush=True)
    print("ring timeouts:", fn["ep24_conv_ring_timeouts"]())


main()
