"""GPU box helper: the three BatchNorm passes on the step's main shapes, launches replayed from a hipGraph over ROTATING buffer sets
(operands not in cache, as in the step).  Run once per library (EP24_LIB=...) on one box to compare builds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
SHAPES = [(32000, 256), (128000, 128), (8000, 512), (512000, 64), (128000, 256), (32000, 512)]
NSET = 6
REPS = 8


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
    print("%-16s %10s %10s %10s   (us per launch: forward, backward reduce, backward apply; %s)" % ("M,C", "fwd", "reduce", "apply", os.environ.get("EP24_LIB", "libep24.so")))
    for M, C in SHAPES:
        zs = [torch.randn(M, C, device=DEV).to(torch.bfloat16) for _ in range(NSET)]
        ys = [torch.zeros(M, C, device=DEV, dtype=torch.bfloat16) for _ in range(NSET)]
        dys = [torch.randn(M, C, device=DEV).to(torch.bfloat16) for _ in range(NSET)]
        stats = torch.zeros(REPS, 2, C, dtype=torch.int64, device=DEV)
        stats[0, 1] = int(M * 2 ** 20)                      # sum of squares = M (variance 1), sum = 0
        gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        nb = torch.zeros(1, dtype=torch.int64, device=DEV)
        save = torch.zeros(2, C, device=DEV)
        save[1] = 1.0
        sums = torch.zeros(REPS, 2, C, dtype=torch.int64, device=DEV)
        gg, bg = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)

        def fwd(s):
            call("bn_act_fwd", ptr(zs[s]), C, ptr(stats), REPS, ptr(gam), ptr(bet), ptr(rm), ptr(rv), ptr(nb), None, ptr(save), ptr(ys[s]), C, None, 0,
                 M, C, 1e-3, 0.03, 1, stream_ptr())

        def red(s):
            call("bn_act_bwd_reduce", ptr(dys[s]), C, ptr(zs[s]), C, ptr(save), ptr(gam), ptr(bet), ptr(sums), sums.data_ptr() + C * 8, M, C, 1, REPS, stream_ptr())

        def app(s):
            call("bn_act_bwd_apply", ptr(dys[s]), C, ptr(zs[s]), C, ptr(save), ptr(gam), ptr(bet), ptr(sums), sums.data_ptr() + C * 8, ptr(gg), ptr(bg),
                 ptr(ys[s]), C, M, C, 1, REPS, stream_ptr())
        t = [min(graph_time(f) for _ in range(2)) for f in (fwd, red, app)]
        print("%-16s %10.1f %10.1f %10.1f" % ("%d,%d" % (M, C), t[0], t[1], t[2]), flush=True)


main()
