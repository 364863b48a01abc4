"""GPU box helper: the loader / consumer ring's in-kernel stamps for the LAST ring launch of a real training step (operands as cold as
the step leaves them, the weight-gradient lane running beside it) - tools/ring_stamps.py measures the same kernel on hot operands alone.
Needs the diagnostic library: EP24_LIB=.../libep24_stamps.so ring_stamps_step.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24 import _lib, loss as eloss, nn as enn, synth, train as etrain  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    model.head.initialize_biases(1e-2)
    model.to(dev)
    ts = etrain.TrainStep(model, eloss.Loss_Function(80), lr=0.001, momentum=0.9, batch=20, size=640)
    images = synth.make_images(20, 640, seed=1).to(dev)
    labels = synth.make_labels(20, 10, size=640, seed=1000).to(dev)
    ts.step(images, labels)
    for _ in range(8):
        ts.step()
    torch.cuda.synchronize()
    L = _lib.lib()
    rd = L.cdll.ep24_debug_read_ring_stamps
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 512)()
    assert rd(buf, 512) == 0
    print("steps  prologue   loop  epilogue | cycles per step  of which waiting for FULL  steps that waited | clock GHz | kernel us")
    for which, name in ((0, "consumer 0"), (1, "consumer 3")):
        rows = [buf[(i * 2 + which) * 8:(i * 2 + which + 1) * 8] for i in range(32)]
        med = [sorted(r[k] for r in rows)[16] for k in range(8)]
        n = max(med[7], 1)
        print("%5d %9d %6d %9d | %15.0f %25.0f %18d | %9.2f | %9.1f  %s" % (n, med[0], med[1], med[2], med[1] / n, med[3] / n, med[4],
                                                                          med[6] / max(med[5], 1) * 0.1, med[5] / 100.0, name))
    print("ring timeouts:", L.fn["ep24_conv_ring_timeouts"]())


main()
