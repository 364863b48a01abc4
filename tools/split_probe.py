"""GPU box probe (timing only, values are not checked): does running the two HALVES of the batch as two concurrent chains hide the
per-launch fixed costs?  A chain of L (3x3 conv -> BatchNorm+SiLU) units at one level, (a) whole batch on one stream, (b) two
half-batch chains on two streams that meet only where BatchNorm needs both halves' statistics (conv A, conv B -> bn A, bn B; the next
conv of a half waits only for its own bn).  Replayed from a hipGraph.  usage: split_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24._lib import call, ptr  # noqa: E402

DEV = "cuda:0"
L = 12


def sp(s):
    return s.cuda_stream


def build(B, H, C, split):
    M = B * H * H
    x = [torch.randn(M, C, device=DEV).to(torch.bfloat16) for _ in range(2)]
    z = torch.zeros(M, C, device=DEV, dtype=torch.bfloat16)
    w = (torch.randn(C, 9, C, device=DEV) * 0.02).to(torch.bfloat16)
    stats = torch.zeros(8, 2, C, dtype=torch.int64, device=DEV)
    gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nb = torch.zeros(1, dtype=torch.int64, device=DEV)
    save = torch.zeros(2, C, device=DEV)

    def conv(src, dst, r0, b, s):
        off = r0 * C * 2
        call("conv_fwd_bf16", src.data_ptr() + off, C, ptr(w), dst.data_ptr() + off, C, 0, 0, 0, None, ptr(stats), 8, b, H, H, C, C, 3, 1, sp(s))

    def bn(src, dst, r0, rows, s):
        off = r0 * C * 2
        call("bn_act_fwd", src.data_ptr() + off, C, ptr(stats), 8, ptr(gam), ptr(bet), ptr(rm), ptr(rv), ptr(nb), None, ptr(save),
             dst.data_ptr() + off, C, None, 0, rows, C, 1e-3, 0.03, 1, sp(s))

    s0 = torch.cuda.Stream()
    s1 = torch.cuda.Stream()

    def body():
        cur = 0
        if not split:
            for _ in range(L):
                conv(x[cur], z, 0, B, s0)
                bn(z, x[1 - cur], 0, M, s0)
                cur = 1 - cur
            return
        hb = B // 2
        hm = hb * H * H
        s1.wait_stream(s0)
        for _ in range(L):
            conv(x[cur], z, 0, hb, s0)
            conv(x[cur], z, hm, B - hb, s1)
            e0, e1 = torch.cuda.Event(), torch.cuda.Event()
            e0.record(s0)
            e1.record(s1)
            s0.wait_event(e1)
            s1.wait_event(e0)
            bn(z, x[1 - cur], 0, hm, s0)
            bn(z, x[1 - cur], hm, M - hm, s1)
            cur = 1 - cur
        s0.wait_stream(s1)

    with torch.cuda.stream(s0):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s0):
        body()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / L)
    return sorted(ts)[3]


print("level (B,H,C)        whole batch, one chain     two half-batch chains    (us per conv + BatchNorm unit, %d units)" % L)
for B, H, C in ((20, 40, 256), (20, 80, 128), (20, 20, 512), (20, 160, 64)):
    a = build(B, H, C, False)
    b = build(B, H, C, True)
    print("%-20s %18.1f %24.1f" % ("%d,%d,%d" % (B, H, C), a, b), flush=True)
