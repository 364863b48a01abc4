"""GPU box helper: time / profile one conv shape through the C ABI.  usage: conv_probe.py kind B H Cin Cout k s [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch
from ep24._lib import call, ptr, stream_ptr

def main():
    kind = sys.argv[1]
    B, H, Cin, Cout, k, s = [int(v) for v in sys.argv[2:8]]
    iters = int(sys.argv[8]) if len(sys.argv) > 8 else 50
    W = H
    OH = (H - 1) // s + 1
    dev = "cuda:0"
    x = torch.randn(B * H * W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, k * k, Cin, device=dev) * 0.05).to(torch.bfloat16)
    wd = (torch.randn(Cin, k * k, Cout, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.zeros(B * OH * OH, Cout, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(B * OH * OH, Cout, device=dev).to(torch.bfloat16)
    dx = torch.zeros(B * H * W, Cin, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(Cout, k * k, Cin, device=dev)
    stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=dev)
    slab = torch.zeros(128 * dw.numel(), device=dev) if (kind == "wgrad" and os.environ.get("EP24_PROBE_SLAB")) else None
    def run():
        if kind == "fwd":
            call("conv_fwd_bf16", ptr(x), Cin, ptr(w), ptr(y), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, k, s, stream_ptr())
        elif kind == "dgrad":
            call("conv_dgrad_bf16", ptr(dy), Cout, ptr(wd), ptr(dx), Cin, 0, B, H, W, Cin, Cout, k, s, stream_ptr())
        elif os.environ.get("EP24_PROBE_SLAB"):
            call("conv_wgrad_slab_bf16", ptr(x), Cin, ptr(dy), Cout, ptr(slab), slab.numel(), k * k * Cin, Cout, Cin, B, H, W, Cin, Cout, k, s, stream_ptr())
        else:
            call("conv_wgrad_bf16", ptr(x), Cin, ptr(dy), Cout, ptr(dw), k * k * Cin, Cout, Cin, B, H, W, Cin, Cout, k, s, stream_ptr())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    # capture the launches in a hipGraph so that the host launch path does not bound short kernels
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        for _ in range(iters):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / iters
    fl = 2.0 * B * OH * OH * Cin * Cout * k * k
    print("%s B%d H%d Cin%d Cout%d k%d s%d: %.1f us  %.1f TFLOP/s" % (kind, B, H, Cin, Cout, k, s, dt * 1e6, fl / dt / 1e12))

main()
