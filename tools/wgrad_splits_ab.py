"""GPU box helper: weight gradient (slab form) + its reduce launch on the 64 x 64-tile layer shapes of YOLOX-l (B = 20), operands rotated over
enough buffer sets to exceed the Infinity Cache, hipGraph-replayed.  Run once per library (EP24_LIB=...) to compare split policies.
usage: wgrad_splits_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch
from ep24 import _lib
from ep24._lib import call, ptr, stream_ptr
DEV = "cuda:0"
SHAPES = [(20, 40, 256, 256, 1, 1, 15), (20, 80, 128, 128, 1, 1, 12), (20, 80, 256, 256, 1, 1, 4), (20, 160, 64, 64, 1, 1, 6), (20, 160, 128, 128, 1, 1, 1),
          (20, 20, 512, 512, 1, 1, 6), (20, 160, 64, 64, 3, 1, 3), (20, 40, 512, 512, 1, 1, 5), (20, 40, 512, 256, 1, 1, 2)]
tot = 0.0
print("lib", os.environ.get("EP24_LIB", "default"))
for B, H, Cin, Cout, k, s, n_in_step in SHAPES:
    M = B * H * H
    nset = max(2, min(8, int(600e6 // (M * (Cin + Cout) * 2)) + 1))
    xs = [torch.randn(M, Cin, device=DEV).to(torch.bfloat16) for _ in range(nset)]
    dys = [torch.randn(M, Cout, device=DEV).to(torch.bfloat16) for _ in range(nset)]
    splits = _lib.lib().fn["ep24_conv_wgrad_splits"](B, H, H, Cin, Cout, k, s)
    numel = Cout * k * k * Cin
    slab = torch.zeros(splits * numel, device=DEV)
    grad = torch.zeros(numel, device=DEV)
    desc = torch.tensor([[0, numel, splits, 0]], dtype=torch.int64, device=DEV)
    st = stream_ptr

    def run():
        for i in range(nset):
            call("conv_wgrad_slab_bf16", ptr(xs[i]), Cin, ptr(dys[i]), Cout, ptr(slab), slab.numel(), k * k * Cin, Cout, Cin, B, H, H, Cin, Cout, k, s, st())
            call("wgrad_reduce", ptr(desc), 1, numel, ptr(grad), ptr(slab), st())

    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            run()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (4 * nset))
    t = sorted(ts)[2]
    tot += t * n_in_step
    print("%-26s splits %4d  wgrad+reduce %7.1f us  x %2d per step" % ("%d,%d,%d,%d,%d,%d" % (B, H, Cin, Cout, k, s), splits, t, n_in_step), flush=True)
print("weighted sum per step: %.1f us" % tot)
