"""GPU box helper: the 1x1 layers of the streaming kernel, launches replayed from a hipGraph over ROTATING buffer sets (the operands
of a launch are not in cache, as in the training step).  Run it once per library (EP24_LIB=...) on one box to compare builds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24 import _lib  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
SHAPES = [(20, 80, 256, 256), (20, 80, 128, 128), (20, 160, 128, 128), (20, 160, 64, 64), (20, 80, 256, 512), (20, 320, 112, 64)]
NSET = 6


def graph_time(run, iters=NSET * 4):
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
    fn = _lib.lib().fn
    print("%-8s %-22s %10s   (us per launch, operands rotated over %d sets; %s)" % ("kind", "B,H,Cin,Cout", "us", NSET, os.environ.get("EP24_LIB", "libep24.so")))
    for B, H, Cin, Cout in SHAPES:
        W = H
        M = B * H * W
        xs = [torch.randn(M, Cin, device=DEV).to(torch.bfloat16) for _ in range(NSET)]
        ys = [torch.zeros(M, Cout, device=DEV, dtype=torch.bfloat16) for _ in range(NSET)]
        dys = [torch.randn(M, Cout, device=DEV).to(torch.bfloat16) for _ in range(NSET)]
        dxs = [torch.zeros(M, Cin, device=DEV, dtype=torch.bfloat16) for _ in range(NSET)]
        w = (torch.randn(Cout, 1, Cin, device=DEV) * 0.05).to(torch.bfloat16)
        wd = (torch.randn(Cin, 1, Cout, device=DEV) * 0.05).to(torch.bfloat16)
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        for kind in ("fwd", "dgrad"):
            def run(s):
                if kind == "fwd":
                    call("conv_fwd_bf16", ptr(xs[s]), Cin, ptr(w), ptr(ys[s]), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, 1, 1, stream_ptr())
                else:
                    call("conv_dgrad_bf16", ptr(dys[s]), Cout, ptr(wd), ptr(dxs[s]), Cin, 0, B, H, W, Cin, Cout, 1, 1, stream_ptr())
            t = min(graph_time(run) for _ in range(2))
            mb = M * (Cin + Cout) * 2 / 1e6
            print("%-8s %-22s %10.1f   %.2f TB/s" % (kind, "%d,%d,%d,%d" % (B, H, Cin, Cout), t, mb / t), flush=True)


main()
