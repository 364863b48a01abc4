"""GPU box helper (diagnostic library: make -C csrc stamps; EP24_LIB=.../libep24_stamps.so): the shader clock the chip holds
INSIDE the training step, from the s_memtime / s_memrealtime stamps of the last halo-patch launch of a step, next to the
same layer launched alone back to back."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ep24 import _lib, loss as eloss, nn as enn, train as etrain, synth  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = torch.device("cuda", 0)
rd = _lib.lib().cdll.ep24_debug_read_stamps
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]


def read():
    buf = (ctypes.c_ulonglong * 512)()
    assert rd(buf, 512) == 0
    rows = [buf[i * 16:i * 16 + 8] for i in range(32)]        # wave 0 of the first 32 workgroups
    med = [sorted(r[k] for r in rows)[16] for k in range(8)]
    return med[6] / max(med[5], 1) * 0.1, med[1] / max(med[7], 1), med[7], med[0], med[2], med[3] / max(med[7], 1), med[4] / max(med[7], 1), med[5] / 100.0


torch.manual_seed(0)
m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
m.head.initialize_biases(1e-2)
m.to(DEV)
ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=0.001, momentum=0.9, batch=20, size=640)
ts.eng.images.copy_(synth.make_images(20, 640, seed=1).to(DEV))
ts.labels.copy_(synth.make_labels(20, 10, size=640, seed=1000).to(DEV))
for i in range(120):
    ts.step()
    if i % 40 == 39:
        torch.cuda.synchronize()
        clk, cyc, n, pro, epi, bar, dma, us = read()
        print("in the step (after %3d steps): clock %.2f GHz, %4.0f cycles per K step (barrier %3.0f, DMA wait %3.0f), %d steps; prologue %5d epilogue %5d cycles; kernel %.1f us" % (
            i + 1, clk, cyc, bar, dma, n, pro, epi, us), flush=True)
del ts, m
torch.cuda.empty_cache()
for B, H, Cin, Cout in [(20, 80, 128, 128), (20, 40, 256, 256), (20, 80, 256, 256)]:
    W = H
    x = torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(Cout, 9, Cin, device=DEV) * 0.05).to(torch.bfloat16)
    y = torch.zeros(B * H * W, Cout, device=DEV, dtype=torch.bfloat16)
    stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
    for _ in range(20000 if H == 40 else 8000):
        call("conv_fwd_bf16", ptr(x), Cin, ptr(w), ptr(y), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, 3, 1, stream_ptr())
    torch.cuda.synchronize()
    clk, cyc, n, pro, epi, bar, dma, us = read()
    print("alone, back to back %-16s: clock %.2f GHz, %4.0f cycles per K step (barrier %3.0f, DMA wait %3.0f), %d steps; prologue %5d epilogue %5d cycles; kernel %.1f us" % (
        "%d,%d,%d,%d" % (B, H, Cin, Cout), clk, cyc, bar, dma, n, pro, epi, us), flush=True)
