"""GPU box helper: stride-2 input gradients as ONE merged launch of the tiled kernel (the default) against one launch per parity class
(kernel_opts bit 2), one process, interleaved rounds, hipGraph replay over rotating operand sets; checks bit-equality."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24 import _lib  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"
SHAPES = [(20, 40, 256, 256), (20, 40, 512, 512), (20, 40, 512, 256), (20, 40, 1024, 512), (20, 20, 512, 512), (20, 20, 1024, 1024),
          (20, 20, 2048, 1024), (20, 20, 1024, 512), (20, 20, 1024, 256), (20, 80, 512, 256), (20, 80, 512, 128)]
NSET = 4


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
    # stride-2 input gradients: four launches (one per parity class, kernel_opts bit 2) against the one merged launch
    print("%-6s %-20s %9s %9s   (us: one launch per parity class, one merged launch)" % ("kind", "B,H,Cin,Cout", "4 x", "merged"))
    tot = [0.0, 0.0]
    for B, H, Cin, Cout in [(20, 320, 64, 128), (20, 160, 128, 256), (20, 80, 256, 512), (20, 40, 512, 1024), (20, 80, 256, 256), (20, 40, 512, 512)]:
        OH = H // 2
        dys = [torch.randn(B * OH * OH, Cout, device=DEV).to(torch.bfloat16) for _ in range(NSET)]
        dxs = [torch.zeros(B * H * H, Cin, device=DEV, dtype=torch.bfloat16) for _ in range(NSET)]
        wd = (torch.randn(Cin, 9, Cout, device=DEV) * 0.05).to(torch.bfloat16)
        ko = [0]

        def run(i):
            call("conv_dgrad_bf16_ex", ptr(dys[i]), Cout, ptr(wd), ptr(dxs[i]), Cin, 0, B, H, H, Cin, Cout, 3, 2, ko[0], stream_ptr())
        outs = {}
        for mode in (4, 0):
            ko[0] = mode
            dxs[0].fill_(7.0)
            run(0)
            torch.cuda.synchronize()
            outs[mode] = dxs[0].clone()
        assert torch.equal(outs[4], outs[0]), (B, H, Cin, Cout)
        res = {4: [], 0: []}
        for rnd in range(2):
            for mode in (4, 0):
                ko[0] = mode
                res[mode].append(graph_time(run))
        t = [min(res[4]), min(res[0])]
        tot[0] += t[0]; tot[1] += t[1]
        print("%-6s %-20s %9.1f %9.1f" % ("dgrad2", "%d,%d,%d,%d" % (B, H, Cin, Cout), t[0], t[1]), flush=True)
    print("sum: four launches %.1f merged %.1f" % tuple(tot))


main()
