"""GPU box helper (round 5): the stride-2 3x3 layers of YOLOX-l (B = 20) - forward and input gradient - with a list of kernel_opts
side by side, interleaved rounds in ONE process, launches replayed from a hipGraph, operands rotated over EP24_AB_SETS sets (default 6:
a launch finds its operands as cold as in the step).
usage: s2_ab.py [opts ...]     (default: 0 and 4 = the one-launch input gradient against one launch per parity class; round 5 used it for the
interleaved class orders recorded in profiles/r05_s2_ab.txt, which were removed again)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
# (B, H_in, Cin, Cout) of the six stride-2 layers: dark2..dark5 downsamples, bu_conv2, bu_conv1
SHAPES = [(20, 320, 64, 128), (20, 160, 128, 256), (20, 80, 256, 512), (20, 40, 512, 1024), (20, 80, 256, 256), (20, 40, 512, 512)]
NSET = max(1, int(os.environ.get("EP24_AB_SETS", "6")))


def graph_time(run, iters=12):
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
    opts = [int(v) for v in sys.argv[1:]] or [0, 4]
    print("# stride-2 3x3 layers, B = 20, %d operand sets; us per launch (best of 3 interleaved rounds), TFLOP/s of the first column" % NSET)
    print("%-6s %-20s " % ("kind", "B,H,Cin,Cout") + " ".join("%10s" % ("opts=%d" % o) for o in opts))
    for B, H, Cin, Cout in SHAPES:
        OH = H // 2
        xs = [torch.randn(B * H * H, Cin, device=DEV).to(torch.bfloat16) for _ in range(NSET)]
        ws = [(torch.randn(Cout, 9, Cin, device=DEV) * 0.05).to(torch.bfloat16) for _ in range(NSET)]
        wds = [(torch.randn(Cin, 9, Cout, device=DEV) * 0.05).to(torch.bfloat16) for _ in range(NSET)]
        ys = [torch.zeros(B * OH * OH, Cout, device=DEV, dtype=torch.bfloat16) for _ in range(NSET)]
        dys = [torch.randn(B * OH * OH, Cout, device=DEV).to(torch.bfloat16) for _ in range(NSET)]
        dxs = [torch.zeros(B * H * H, Cin, device=DEV, dtype=torch.bfloat16) for _ in range(NSET)]
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        flops = 2.0 * B * OH * OH * Cout * 9 * Cin
        for kind in ("fwd", "dgrad"):
            best = {o: 1e9 for o in opts}
            for _ in range(3):
                for o in opts:
                    if kind == "fwd":
                        run = lambda i, o=o: call("conv_fwd_bf16_ex", ptr(xs[i]), Cin, ptr(ws[i]), ptr(ys[i]), Cout, 0, 0, 0, None, ptr(stats), 8,
                                                  B, H, H, Cin, Cout, 3, 2, o, stream_ptr())
                    else:
                        run = lambda i, o=o: call("conv_dgrad_bf16_ex", ptr(dys[i]), Cout, ptr(wds[i]), ptr(dxs[i]), Cin, 0, B, H, H, Cin, Cout, 3, 2, o,
                                                  stream_ptr())
                    best[o] = min(best[o], graph_time(run))
            print("%-6s %-20s " % (kind, "%d,%d,%d,%d" % (B, H, Cin, Cout)) + " ".join("%10.1f" % best[o] for o in opts) +
                  "   TF: " + " ".join("%5.0f" % (flops / best[o] / 1e6) for o in opts))
        del xs, ws, wds, ys, dys, dxs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
