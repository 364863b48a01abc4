"""GPU box helper: the transformed-A forms of the streaming 1x1 kernel (ep24_conv1x1_bnin_bf16, ep24_conv1x1_dgrad_bnbwd_bf16) against
the two launches each replaces, replayed from a hipGraph over ROTATING operand sets (cold operands, as in the step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16
SHAPES = [(20, 80, 128, 128), (20, 160, 64, 64), (20, 40, 128, 128)]
NSET = 6
R = 8


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
    print("us per unit (operands rotated over %d sets; %s)" % (NSET, os.environ.get("EP24_LIB", "libep24.so")))
    for B, H, Cin, Cout in SHAPES:
        W = H
        M = B * H * W
        zs = [torch.randn(M, Cin, device=DEV).to(BF) for _ in range(NSET)]
        rs = [torch.randn(M, Cin, device=DEV).to(BF) for _ in range(NSET)]
        ys = [torch.zeros(M, Cin, device=DEV, dtype=BF) for _ in range(NSET)]
        os_ = [torch.zeros(M, Cout, device=DEV, dtype=BF) for _ in range(NSET)]
        w = (torch.randn(Cout, 1, Cin, device=DEV) * 0.05).to(BF)
        gam, bet = torch.rand(Cin, device=DEV) + 0.5, torch.rand(Cin, device=DEV) - 0.5
        save = torch.zeros(2, Cin, device=DEV)
        stats_in = torch.zeros(R, 2, Cin, dtype=torch.int64, device=DEV)
        stats_in[0, 1] = M << 20
        stats_out = torch.zeros(R, 2, Cout, dtype=torch.int64, device=DEV)
        sp = stream_ptr
        for res in (True, False):
            def bn(s):
                call("bn_act_fwd", ptr(zs[s]), Cin, ptr(stats_in), R, ptr(gam), ptr(bet), None, None, None, None, ptr(save), ptr(ys[s]), Cin,
                     ptr(rs[s]) if res else None, Cin if res else 0, M, Cin, 1e-3, 0.03, 1, sp())

            def conv(s):
                call("conv_fwd_bf16", ptr(ys[s]), Cin, ptr(w), ptr(os_[s]), Cout, 0, 0, 0, None, ptr(stats_out), R, B, H, W, Cin, Cout, 1, 1, sp())

            def fused(s):
                call("conv1x1_bnin_bf16", ptr(zs[s]), Cin, ptr(stats_in), R, ptr(gam), ptr(bet), None, None, None, None, ptr(save), ptr(ys[s]), Cin,
                     ptr(rs[s]) if res else None, Cin if res else 0, 1e-3, 0.03, 1, ptr(w), ptr(os_[s]), Cout, ptr(stats_out), R, B, H, W, Cin, Cout, sp())
            t_bn, t_conv, t_both, t_f = graph_time(bn), graph_time(conv), graph_time(lambda s: (bn(s), conv(s))), graph_time(fused)
            print("fwd  %d,%d,%d->%d res=%d : bn %.1f + conv %.1f = %.1f (back to back %.1f) | fused %.1f" % (B, H, Cin, Cout, res, t_bn, t_conv, t_bn + t_conv, t_both, t_f), flush=True)
        # backward: this unit is the 1x1 conv Cin -> Cout; dy / z / dz have Cout channels, dx has Cin
        dys = [torch.randn(M, Cout, device=DEV).to(BF) for _ in range(NSET)]
        z2 = [torch.randn(M, Cout, device=DEV).to(BF) for _ in range(NSET)]
        dzs = [torch.zeros(M, Cout, device=DEV, dtype=BF) for _ in range(NSET)]
        dxs = [torch.zeros(M, Cin, device=DEV, dtype=BF) for _ in range(NSET)]
        wd = (torch.randn(Cin, 1, Cout, device=DEV) * 0.05).to(BF)
        g2, b2 = torch.rand(Cout, device=DEV) + 0.5, torch.rand(Cout, device=DEV) - 0.5
        sv2 = torch.zeros(2, Cout, device=DEV)
        sv2[1] = 1.0
        sums = torch.zeros(R, 2, Cout, dtype=torch.int64, device=DEV)
        gg, bg = torch.zeros(Cout, device=DEV), torch.zeros(Cout, device=DEV)
        for acc in (1, 0):
            def ap(s):
                call("bn_act_bwd_apply", ptr(dys[s]), Cout, ptr(z2[s]), Cout, ptr(sv2), ptr(g2), ptr(b2), ptr(sums), ptr(sums, Cout), ptr(gg), ptr(bg),
                     ptr(dzs[s]), Cout, M, Cout, 1, R, sp())

            def dg(s):
                call("conv_dgrad_bf16", ptr(dzs[s]), Cout, ptr(wd), ptr(dxs[s]), Cin, acc, B, H, W, Cin, Cout, 1, 1, sp())

            def fb(s):
                call("conv1x1_dgrad_bnbwd_bf16", ptr(dys[s]), Cout, ptr(z2[s]), Cout, ptr(sv2), ptr(g2), ptr(b2), ptr(sums), ptr(sums, Cout), ptr(gg), ptr(bg),
                     ptr(dzs[s]), Cout, 1, R, ptr(wd), ptr(dxs[s]), Cin, acc, B, H, W, Cin, Cout, sp())
            t_ap, t_dg, t_both, t_f = graph_time(ap), graph_time(dg), graph_time(lambda s: (ap(s), dg(s))), graph_time(fb)
            print("bwd  %d,%d,%d->%d acc=%d : apply %.1f + dgrad %.1f = %.1f (back to back %.1f) | fused %.1f" % (B, H, Cin, Cout, acc, t_ap, t_dg, t_ap + t_dg, t_both, t_f), flush=True)


if __name__ == "__main__":
    main()
