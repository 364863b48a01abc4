"""GPU box helper (round 5): the BatchNorm backward of a unit as two launches (reduce, apply) against the one-launch form with a grid-wide
wait (ep24_bn_act_bwd_fused); operands rotated over 6 buffer sets, launches replayed from a hipGraph, us per unit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch
from ep24._lib import call, ptr, stream_ptr as sp
DEV, BF, NSET, R = "cuda:0", torch.bfloat16, 6, 8
print("# M,C   two launches (reduce + apply)   one launch   (us per unit, median of 5 replays of 24 units)")
for M, C in [(32000, 256), (128000, 128), (8000, 512), (512000, 64), (128000, 256), (32000, 512), (8000, 1024)]:
    zs = [torch.randn(M, C, device=DEV).to(BF) for _ in range(NSET)]
    dys = [torch.randn(M, C, device=DEV).to(BF) for _ in range(NSET)]
    dzs = [torch.zeros(M, C, dtype=BF, device=DEV) for _ in range(NSET)]
    save = torch.zeros(2, C, device=DEV); save[1] = 1.0
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    gg, bg = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    sums = torch.zeros(R, 2, C, dtype=torch.int64, device=DEV)
    bar = torch.zeros(64, dtype=torch.int32, device=DEV)

    def two(i):
        call("memset_zero", ptr(sums), sums.numel() * 8, sp())
        call("bn_act_bwd_reduce", ptr(dys[i]), C, ptr(zs[i]), C, ptr(save), ptr(g), ptr(b), ptr(sums), ptr(sums, C), M, C, 1, R, sp())
        call("bn_act_bwd_apply", ptr(dys[i]), C, ptr(zs[i]), C, ptr(save), ptr(g), ptr(b), ptr(sums), ptr(sums, C), ptr(gg), ptr(bg), ptr(dzs[i]), C, M, C, 1, R, sp())

    def one(i):
        call("memset_zero", ptr(sums), sums.numel() * 8, sp())
        call("memset_zero", ptr(bar), 8, sp())
        call("bn_act_bwd_fused", ptr(dys[i]), C, ptr(zs[i]), C, ptr(save), ptr(g), ptr(b), ptr(sums), ptr(sums, C), ptr(gg), ptr(bg), ptr(dzs[i]), C, M, C, 1, R, ptr(bar), sp())

    res = []
    for fn in (two, one):
        for i in range(NSET):
            fn(i)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(24):
                fn(i % NSET)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 24)
        res.append(sorted(ts)[2])
    print("%7d,%-5d %10.1f %22.1f" % (M, C, res[0], res[1]))
    del zs, dys, dzs
    torch.cuda.empty_cache()
