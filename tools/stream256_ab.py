"""GPU box helper: 1x1 stride-1 layers with 128 < K <= 256 and M < 100 000 in the tiled kernel (kernel_opts bit 8: the default before round 5) against the
streaming kernel (the default), forward and input gradient, hipGraph replay over rotating operand sets."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402
from xf_ab import graph_time, NSET  # noqa: E402

DEV, BF = "cuda:0", torch.bfloat16
SHAPES = [(20, 40, 256, 256), (20, 20, 256, 256), (20, 40, 256, 128), (20, 40, 256, 512), (20, 40, 192, 256), (20, 80, 256, 256)]


def main():
    for B, H, Cin, Cout in SHAPES:
        W, M = H, B * H * H
        xs = [torch.randn(M, Cin, device=DEV).to(BF) for _ in range(NSET)]
        ys = [torch.zeros(M, Cout, device=DEV, dtype=BF) for _ in range(NSET)]
        dys = [torch.randn(M, Cout, device=DEV).to(BF) for _ in range(NSET)]
        dxs = [torch.zeros(M, Cin, device=DEV, dtype=BF) for _ in range(NSET)]
        w = (torch.randn(Cout, 1, Cin, device=DEV) * 0.05).to(BF)
        wd = (torch.randn(Cin, 1, Cout, device=DEV) * 0.05).to(BF)
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        out = []
        for ko in (256, 0):
            f = graph_time(lambda s: call("conv_fwd_bf16_ex", ptr(xs[s]), Cin, ptr(w), ptr(ys[s]), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, 1, 1, ko, stream_ptr()))
            d = graph_time(lambda s: call("conv_dgrad_bf16_ex", ptr(dys[s]), Cout, ptr(wd), ptr(dxs[s]), Cin, 1, B, H, W, Cin, Cout, 1, 1, ko, stream_ptr())) if Cout <= 256 else float("nan")
            out.append((f, d))
        print("%d,%d,%d->%d : fwd tiled %.1f stream %.1f | dgrad(acc) tiled %.1f stream %.1f" % (B, H, Cin, Cout, out[0][0], out[1][0], out[0][1], out[1][1]), flush=True)


if __name__ == "__main__":
    main()
