"""GPU box helper: the 3x3 stride-1 layers with <= 64 channels in the weights-in-registers kernel (conv_wreg.hip, the default) against
the tiled kernel (kernel_opts bit 9), launches replayed from a hipGraph, EP24_AB_SETS operand sets (cold operands), interleaved rounds.
usage: wreg_ab.py ["B,H,Cin,Cout;..."]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24 import _lib  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
SHAPES = [(20, 160, 64, 64), (4, 160, 64, 64), (20, 160, 48, 48), (20, 80, 64, 64)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in sys.argv[1].split(";")]
NSET = max(1, int(os.environ.get("EP24_AB_SETS", "4")))
KOS = tuple(int(v) for v in os.environ.get("EP24_AB_KOS", "512,0").split(","))      # kernel_opts compared (one value: the loaded library's default kernel alone)


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
    fn = _lib.lib().fn
    print("%-6s %-18s %10s %10s   (us per launch: tiled kernel, weights-in-registers kernel; TFLOP/s)" % ("kind", "B,H,Cin,Cout", "tiled", "wreg"))
    for shp in SHAPES:
        B, H, Cin, Cout = shp[:4]
        k_, s_ = (shp[4], shp[5]) if len(shp) > 5 else (3, 1)     # "B,H,Cin,Cout[,k,s]"
        W = H
        OH = (H + 2 * ((k_ - 1) // 2) - k_) // s_ + 1
        xs = [torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16) for _ in range(NSET)]
        ws = [(torch.randn(Cout, k_ * k_, Cin, device=DEV) * 0.05).to(torch.bfloat16) for _ in range(NSET)]
        wds = [(torch.randn(Cin, k_ * k_, Cout, device=DEV) * 0.05).to(torch.bfloat16) for _ in range(NSET)]
        ys = [torch.zeros(B * OH * OH, Cout, device=DEV, dtype=torch.bfloat16) for _ in range(NSET)]
        dys = [torch.randn(B * OH * OH, Cout, device=DEV).to(torch.bfloat16) for _ in range(NSET)]
        dxs = [torch.zeros(B * H * W, Cin, device=DEV, dtype=torch.bfloat16) for _ in range(NSET)]
        st = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        kid = fn["ep24_conv_kernel_for"](0, B, H, W, Cin, Cout, k_, s_, 0, 0)
        for kind in ("fwd", "dgrad"):
            best = {}
            for rnd in range(3):
                for ko in KOS:
                    if kind == "fwd":
                        run = lambda i, ko=ko: call("conv_fwd_bf16_ex", ptr(xs[i]), Cin, ptr(ws[i]), ptr(ys[i]), Cout, 0, 0, 0, None, ptr(st), 8, B, H, W, Cin, Cout, k_, s_, ko, stream_ptr())
                    else:
                        run = lambda i, ko=ko: call("conv_dgrad_bf16_ex", ptr(dys[i]), Cout, ptr(wds[i]), ptr(dxs[i]), Cin, 0, B, H, W, Cin, Cout, k_, s_, ko, stream_ptr())
                    t = graph_time(run)
                    best[ko] = min(best.get(ko, 1e9), t)
            gf = 2.0 * B * OH * OH * k_ * k_ * Cin * Cout / 1e9
            print("%-6s %-22s " % (kind, ",".join(str(v) for v in shp)) + " ".join("%10.1f" % best[k] for k in KOS) + "   TF: " + " ".join("%6.0f" % (gf / best[k] * 1e-3) for k in KOS) + "   (default kernel id %d)" % kid, flush=True)
    print("ring timeouts:", fn["ep24_conv_ring_timeouts"]())


if __name__ == "__main__":
    main()
