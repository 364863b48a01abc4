"""GPU box helper: LDS ring depth A/B (2 = two-stage loop, 3, 4) of the tiled conv kernel (forward / input gradient) and of the
weight-gradient kernel on the YOLOX-l layer shapes at B = 20, in ONE process, interleaved rounds, launches replayed from a hipGraph
over ROTATING buffer sets (operands not in cache, as in the step).  Checks on the way that every depth gives bit-identical results.
usage: ring_ab.py [fwd dgrad wgrad]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24 import _lib  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
# (B, H, Cin, Cout, k, s): what the tiled kernel keeps (1x1 with K > 128, the 20x20 / 160x160 3x3 layers, stride 2) + every wgrad class
SHAPES = [(20, 40, 256, 256, 1, 1), (20, 40, 512, 512, 1, 1), (20, 40, 512, 256, 1, 1), (20, 40, 1024, 512, 1, 1),
          (20, 20, 512, 512, 1, 1), (20, 20, 1024, 1024, 1, 1), (20, 20, 2048, 1024, 1, 1), (20, 20, 1024, 512, 1, 1), (20, 80, 512, 256, 1, 1),
          (20, 20, 512, 512, 3, 1), (20, 160, 64, 64, 3, 1), (20, 20, 256, 256, 3, 1),
          (20, 320, 64, 128, 3, 2), (20, 160, 128, 256, 3, 2), (20, 80, 256, 512, 3, 2), (20, 40, 512, 1024, 3, 2),
          (20, 80, 128, 128, 1, 1), (20, 80, 256, 256, 1, 1), (20, 160, 64, 64, 1, 1), (20, 160, 128, 128, 1, 1),
          (20, 40, 256, 256, 3, 1), (20, 80, 128, 128, 3, 1), (20, 80, 256, 256, 3, 1)]
NSET = 4
MODES = (16, 4, 8)           # kernel_opts: two-stage loop, 3 stages, 4 stages


def graph_time(run, iters=NSET * 3):
    for s in range(NSET):
        run(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            run(i % NSET)
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
    kinds = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
    fn = _lib.lib().fn
    print("%-6s %-24s %9s %9s %9s   (us per launch: ring depth 2, 3, 4; operands rotated over %d sets)  kernel" % ("kind", "B,H,Cin,Cout,k,s", "ns2", "ns3", "ns4", NSET))
    tot = {}
    for B, H, Cin, Cout, k, s in SHAPES:
        W = H
        OH = (H - 1) // s + 1
        M, MO = B * H * W, B * OH * OH
        nset = NSET if M * max(Cin, Cout) < 200e6 else 2
        xs = [torch.randn(M, Cin, device=DEV).to(torch.bfloat16) for _ in range(nset)]
        dys = [torch.randn(MO, Cout, device=DEV).to(torch.bfloat16) for _ in range(nset)]
        ys = [torch.zeros(MO, Cout, device=DEV, dtype=torch.bfloat16) for _ in range(nset)]
        dxs = [torch.zeros(M, Cin, device=DEV, dtype=torch.bfloat16) for _ in range(nset)]
        w = (torch.randn(Cout, k * k, Cin, device=DEV) * 0.05).to(torch.bfloat16)
        wd = (torch.randn(Cin, k * k, Cout, device=DEV) * 0.05).to(torch.bfloat16)
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        numel = Cout * k * k * Cin
        for kind in kinds:
            if kind != "wgrad":
                kid = fn["ep24_conv_kernel_for"](0 if kind == "fwd" else 1, B, H, W, Cin, Cout, k, s, 0, 0)
                if kid != 0:
                    continue                                  # the patch / streaming kernel has this shape
            ko = [0]
            slabs = {}

            def run(i):
                i %= nset
                if kind == "fwd":
                    call("conv_fwd_bf16_ex", ptr(xs[i]), Cin, ptr(w), ptr(ys[i]), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, k, s, ko[0], stream_ptr())
                elif kind == "dgrad":
                    call("conv_dgrad_bf16_ex", ptr(dys[i]), Cout, ptr(wd), ptr(dxs[i]), Cin, 0, B, H, W, Cin, Cout, k, s, ko[0], stream_ptr())
                else:
                    sl = slabs[ko[0]]
                    call("conv_wgrad_slab_bf16_ex", ptr(xs[i]), Cin, ptr(dys[i]), Cout, ptr(sl), sl.numel(), k * k * Cin, Cout, Cin, B, H, W, Cin, Cout, k, s,
                         ko[0], stream_ptr())

            outs, res = {}, {}
            for mode in MODES:                               # results first: every ring depth must give the same bits
                ko[0] = mode
                if kind == "wgrad":
                    sp = fn["ep24_conv_wgrad_splits_ex"](B, H, W, Cin, Cout, k, s, mode)
                    slabs[mode] = torch.zeros(sp * numel, device=DEV)
                    run(0)
                    torch.cuda.synchronize()
                    outs[mode] = slabs[mode].view(sp, numel).double().sum(0)       # split counts differ with the depth: compare the sums
                else:
                    (ys if kind == "fwd" else dxs)[0].zero_()
                    run(0)
                    torch.cuda.synchronize()
                    outs[mode] = (ys if kind == "fwd" else dxs)[0].clone()
            ref = outs[MODES[0]]
            for mode in MODES[1:]:
                if kind == "wgrad":
                    err = float((outs[mode] - ref).abs().max() / ref.abs().max())
                    assert err < 1e-5, (kind, B, H, Cin, Cout, k, s, mode, err)
                else:
                    assert torch.equal(outs[mode], ref), (kind, B, H, Cin, Cout, k, s, mode)
            for rnd in range(2):
                for mode in MODES:
                    ko[0] = mode
                    res.setdefault(mode, []).append(graph_time(run))
            t = [min(res[m]) for m in MODES]
            for m, v in zip(MODES, t):
                tot[(kind, m)] = tot.get((kind, m), 0.0) + v
            print("%-6s %-24s %9.1f %9.1f %9.1f   best ns%d" % (kind, "%d,%d,%d,%d,%d,%d" % (B, H, Cin, Cout, k, s), t[0], t[1], t[2], 2 + t.index(min(t))), flush=True)
    for kind in kinds:
        print("sum %-6s" % kind, " ".join("%9.1f" % tot.get((kind, m), 0.0) for m in MODES))


main()
