"""GPU box helper: throughput of the input-pipeline kernels, sources resident in HBM and PCIe-inclusive,
next to the CPU oracle.  usage: input_probe.py [batch] [h] [w] [S]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ep24 import input as ein
from ep24._lib import call, ptr, stream_ptr


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 480
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    S = int(sys.argv[4]) if len(sys.argv) > 4 else 640
    dev = torch.device("cuda:0")
    g = np.random.RandomState(0)
    imgs = [g.randint(0, 256, (h, w, 3)).astype(np.uint8) for _ in range(B)]
    tgts = [np.concatenate([g.randint(0, 80, (10, 1)).astype(np.float64), g.rand(10, 50)], 1) for _ in range(B)]
    r, rh, rw = ein.letterbox_geometry(h, w, (S, S))
    src = torch.stack([torch.from_numpy(i) for i in imgs]).to(dev)
    desc = torch.tensor([[i * h * w * 3, h, w, 3 * w, rh, rw] for i in range(B)], dtype=torch.int64, device=dev)
    sc = torch.tensor([[1.0 / (rw / w), 1.0 / (rh / h)]] * B, dtype=torch.float64, device=dev)
    out = torch.empty(B, 3, S, S, device=dev)

    def run():
        call("preproc_u8", ptr(src), ptr(desc), ptr(sc), B, ptr(out), S, S, stream_ptr())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / 50
    by = B * (rh * rw * 3 + S * S * 12)               # resized-area source bytes once + the fp32 canvas
    print("preproc_u8, sources resident: %d x %dx%d -> %dx%d: %.1f us = %.0f images/s, %.2f TB/s of algorithmic bytes (%.1f MB)"
          % (B, h, w, S, S, dt * 1e6, B / dt, by / dt / 1e12, by / 1e6))
    # PCIe-inclusive: pinned raw bytes -> device -> kernels (what a step of the loader costs)
    pinned = [torch.from_numpy(i).pin_memory() for i in imgs]
    tt = ein.TrainTransform()
    oi = torch.empty(B, 3, S, S, device=dev)
    ol = torch.empty(B, 50, 51, device=dev)
    for _ in range(2):
        tt.batch(pinned, tgts, (S, S), oi, ol)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        tt.batch(pinned, tgts, (S, S), oi, ol)
    torch.cuda.synchronize()
    dp = (time.time() - t0) / 10
    print("TrainTransform.batch from pinned host images (upload %.1f MB + 2 launches + host bookkeeping): %.2f ms = %.0f images/s;"
          " the fp32 canvases the reference ships would be %.1f MB" % (B * h * w * 3 / 1e6, dp * 1e3, B / dp, B * S * S * 12 / 1e6))
    from oracle import input as oin
    t0 = time.time()
    for i in range(4):
        oin.train_transform(imgs[i], tgts[i], (S, S))
    dc = (time.time() - t0) / 4
    print("CPU oracle (numpy, 1 thread): %.1f ms per image = %.1f images/s" % (dc * 1e3, 1 / dc))


main()
