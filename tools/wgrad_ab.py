"""GPU box helper: A/B of the weight-gradient ring (wgrad_ring_kernel, kernel_opts bit 0) against wgrad_kernel (the default) on
the layers of YOLOX-l (B = 20) with Cout >= 256, each with its slab reduce.  Interleaved rounds in ONE process, launches replayed from
a hipGraph, operands rotating over several buffer sets.  usage: wgrad_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24 import _lib  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
SHAPES = [(20, 40, 256, 256, 3, 1), (20, 20, 512, 512, 3, 1), (20, 80, 256, 256, 3, 1), (20, 80, 256, 512, 3, 1), (20, 40, 256, 512, 3, 1),
          (20, 80, 256, 512, 3, 2), (20, 40, 512, 1024, 3, 2), (20, 40, 512, 512, 1, 1), (20, 20, 1024, 1024, 1, 1), (20, 20, 2048, 1024, 1, 1),
          (20, 40, 256, 256, 1, 1), (20, 40, 1024, 512, 1, 1)]
SETS = 4


def graph_time(run, iters=12):
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
    print("%-26s %14s %14s   (us per launch incl. its slab reduce: wgrad_kernel, ring; splits; TFLOP/s)" % ("B,H,Cin,Cout,k,s", "wgrad_kernel", "ring"))
    for B, H, Cin, Cout, k, s in SHAPES:
        W = H
        OH = (H - 1) // s + 1
        T = k * k
        xs = [torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16) for _ in range(SETS)]
        dys = [torch.randn(B * OH * OH, Cout, device=DEV).to(torch.bfloat16) for _ in range(SETS)]
        numel = Cout * T * Cin
        fl = 2.0 * B * OH * OH * Cin * Cout * T
        res, sps = {}, {}
        for rnd in range(3):
            for wo in (0, 1):
                splits = fn["ep24_conv_wgrad_splits_ex"](B, H, W, Cin, Cout, k, s, wo)
                sps[wo] = splits
                slab = torch.zeros(splits * numel, device=DEV)
                g = torch.zeros(numel, device=DEV)
                desc = torch.tensor([[0, numel, splits, 0]], dtype=torch.int64, device=DEV)

                def run(i):
                    call("conv_wgrad_slab_bf16_ex", ptr(xs[i]), Cin, ptr(dys[i]), Cout, ptr(slab), splits * numel, T * Cin, Cout, Cin, B, H, W, Cin, Cout, k, s, wo, stream_ptr())
                    call("wgrad_reduce", ptr(desc), 1, numel, ptr(g), ptr(slab), stream_ptr())
                res.setdefault(wo, []).append(graph_time(run))
        print("%-26s %14.1f %14.1f   splits %3d %3d   TF %5.0f %5.0f" % ("%d,%d,%d,%d,%d,%d" % (B, H, Cin, Cout, k, s), min(res[0]), min(res[1]), sps[0], sps[1],
                                                                       fl / min(res[0]) / 1e6, fl / min(res[1]) / 1e6), flush=True)
    print("ring timeouts:", fn["ep24_conv_ring_timeouts"]())


main()
