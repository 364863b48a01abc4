"""GPU box helper: A/B of the three kernels of the 3x3 stride-1 layers of YOLOX-l (B = 20) - generic tiled, 8-wave halo patch,
loader / consumer ring - interleaved rounds in ONE process (median of 5 replays each, best of 3 rounds), launches replayed from a hipGraph.
usage: conv_ab.py [fwd|dgrad ...]"""
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


if os.environ.get("EP24_AB_SHAPES"):                    # "B,H,Cin,Cout;..." (e.g. channel counts whose weight-row stride is not a power of two)
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["EP24_AB_SHAPES"].split(";")]


# EP24_AB_SETS=n (n > 1): every launch of a replay works on the next of n operand sets, so that it finds its operands as cold as a
# launch of the training step does (round 4: the three-stage tiled form wins in the step and loses on one hot set - the answer of an A/B
# depends on what the launch finds in the caches)
NSET = max(1, int(os.environ.get("EP24_AB_SETS", "1")))


def graph_time(run, iters=24):
    for i in range(NSET):
        run(i)
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
    kinds = sys.argv[1:] or ["fwd", "dgrad"]
    fn = _lib.lib().fn
    print("%-8s %-22s %10s %10s %10s %10s   (us: generic tiled kernel, 8-wave halo-patch kernel, loader / consumer ring with 16x16x32 and with 32x32x16 consumers)" % ("kind", "B,H,Cin,Cout", "tiled", "patch8", "ring16", "ring32"))
    for B, H, Cin, Cout in SHAPES:
        W = H
        zero = os.environ.get("EP24_PROBE_ZERO") == "1"          # DVFS check: the same kernels on all-zero operands
        xs = [torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16) * (0 if zero else 1) for _ in range(NSET)]
        ws = [(torch.randn(Cout, 9, Cin, device=DEV) * 0.05).to(torch.bfloat16) * (0 if zero else 1) for _ in range(NSET)]
        wds = [(torch.randn(Cin, 9, Cout, device=DEV) * 0.05).to(torch.bfloat16) * (0 if zero else 1) for _ in range(NSET)]
        ys = [torch.zeros(B * H * W, Cout, device=DEV, dtype=torch.bfloat16) for _ in range(NSET)]
        dys = [torch.randn(B * H * W, Cout, device=DEV).to(torch.bfloat16) * (0 if zero else 1) for _ in range(NSET)]
        dxs = [torch.zeros(B * H * W, Cin, device=DEV, dtype=torch.bfloat16) for _ in range(NSET)]
        z = torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16)
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        save = torch.ones(2, Cin, device=DEV)
        gam, bet = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
        sums = torch.zeros(2, Cin, dtype=torch.int64, device=DEV)
        fl = 2.0 * B * H * W * Cin * Cout * 9
        for kind in kinds:
            ko = [0]                                     # kernel_opts of the _ex entry points: bit 0 tiled kernel, bit 1 narrow epilogue

            def run(s=0):
                if kind == "fwd":
                    call("conv_fwd_bf16_ex", ptr(xs[s]), Cin, ptr(ws[s]), ptr(ys[s]), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, 3, 1, ko[0], stream_ptr())
                elif kind == "dgrad":
                    call("conv_dgrad_bf16_ex", ptr(dys[s]), Cout, ptr(wds[s]), ptr(dxs[s]), Cin, 0, B, H, W, Cin, Cout, 3, 1, ko[0], stream_ptr())
                else:
                    raise SystemExit("kinds: fwd dgrad")
            res = {}
            for rnd in range(3):                         # interleaved rounds in one process
                for mode in (1, 8, 0, 32, 64, 129):           # kernel_opts: bit 0 tiled kernel, bit 3 8-wave halo-patch kernel, 0 the default (ring, 16x16x32), bit 5 ring with 32x32x16, bit 6 narrow ring
                    ko[0] = mode
                    res.setdefault(mode, []).append(graph_time(run))
            print("%-8s %-22s %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f   TF: %5.0f %5.0f %5.0f %5.0f %5.0f %5.0f" % (
                kind, "%d,%d,%d,%d" % (B, H, Cin, Cout), min(res[1]), min(res[8]), min(res[0]), min(res[32]), min(res[64]), min(res[129]),
                fl / min(res[1]) / 1e6, fl / min(res[8]) / 1e6, fl / min(res[0]) / 1e6, fl / min(res[32]) / 1e6, fl / min(res[64]) / 1e6, fl / min(res[129]) / 1e6), flush=True)
    print("ring timeouts:", fn["ep24_conv_ring_timeouts"]())


main()
