"""GPU box helper: time the BN+SiLU streaming kernels per layer shape through the C ABI.

usage: bn_probe.py [M,C ...]   (default: the shapes of YOLOX-l at B=20)
Every launch works on its own buffer set (enough sets to exceed the 256 MB MALL) so the numbers are HBM numbers,
and the launches are replayed from a hipGraph so the host launch path does not bound the short kernels.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch
from ep24._lib import call, ptr, stream_ptr

DEV = "cuda:0"
SHAPES = [(32000, 256), (128000, 128), (8000, 512), (512000, 64), (128000, 256), (2048000, 64), (8000, 1024)]


def probe(M, C, kind, warm):
    bytes_t = M * C * 2
    nset = 1 if warm else max(2, min(24, int(600e6 // (3 * bytes_t)) + 1))
    sets = []
    for i in range(nset):
        z = torch.randn(M, C, device=DEV).to(torch.bfloat16)
        dy = torch.randn(M, C, device=DEV).to(torch.bfloat16)
        out = torch.empty(M, C, device=DEV, dtype=torch.bfloat16)
        sets.append((z, dy, out))
    R = 8                                                   # replicas of the statistics / backward sums, as the engine runs them
    stats = torch.zeros(R, 2, C, dtype=torch.int64, device=DEV)
    zf = sets[0][0].float()
    stats[0, 0] = (zf.sum(0) * (1 << 20)).long()
    stats[0, 1] = ((zf * zf).sum(0) * (1 << 20)).long()
    gamma = torch.ones(C, device=DEV)
    beta = torch.zeros(C, device=DEV)
    save = torch.zeros(2, C, device=DEV)
    save[1] = 1.0
    sums = torch.zeros(R, 2, C, dtype=torch.int64, device=DEV)
    iters = max(nset, 24)

    def run(i):
        z, dy, out = sets[i % nset]
        s = stream_ptr()
        if kind == "fwd":
            call("bn_act_fwd", ptr(z), C, ptr(stats), R, ptr(gamma), ptr(beta), None, None, None, None, ptr(save), ptr(out), C,
                 None, 0, M, C, 1e-3, 0.03, 1, s)
        elif kind == "reduce":
            call("bn_act_bwd_reduce", ptr(dy), C, ptr(z), C, ptr(save), ptr(gamma), ptr(beta), ptr(sums), ptr(sums[0, 1]), M, C, 1, R, s)
        else:
            call("bn_act_bwd_apply", ptr(dy), C, ptr(z), C, ptr(save), ptr(gamma), ptr(beta), ptr(sums), ptr(sums[0, 1]),
                 None, None, ptr(out), C, M, C, 1, R, s)

    run(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            run(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    passes = {"fwd": 2, "reduce": 2, "apply": 3}[kind]
    return us, passes * bytes_t / us / 1e3


def main():
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or SHAPES
    print("%-16s %-7s %10s %10s   %10s %10s" % ("M,C", "kernel", "cold us", "GB/s", "warm us", "GB/s"))   # GB/s of algorithmic bytes
    for M, C in shapes:
        for kind in ("fwd", "reduce", "apply"):
            c = probe(M, C, kind, False)
            w = probe(M, C, kind, True)
            print("%-16s %-7s %10.1f %10.0f   %10.1f %10.0f" % ("%d,%d" % (M, C), kind, c[0], c[1], w[0], w[1]), flush=True)


main()
