"""GPU box helper: throughput of the label-generation kernels with the masks resident in HBM, next to the
CPU oracle.  usage: labels_probe.py [n_objects] [H] [W]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ep24 import labels24
from ep24._lib import call, ptr, stream_ptr


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    g = np.random.RandomState(0)
    yy, xx = np.mgrid[0:H, 0:W]
    base = []
    for _ in range(16):                                        # 16 distinct blobs, repeated: the kernel does not care
        m = np.zeros((H, W), np.uint8)
        for _ in range(4):
            cy, cx, r = g.uniform(H * .3, H * .7), g.uniform(W * .3, W * .7), g.uniform(20, 90)
            m |= (np.hypot(yy - cy, xx - cx) <= r).astype(np.uint8)
        ys, xs = np.nonzero(m)
        base.append((m, (xs.min() + xs.max()) / 2.0, (ys.min() + ys.max()) / 2.0))
    dev = torch.device("cuda:0")
    L = int(np.sqrt(H * H + W * W))
    ns = int(np.ceil(L / 0.2))
    masks = torch.stack([torch.from_numpy(base[i % 16][0]) for i in range(n)]).to(dev)
    desc = torch.tensor([[i * H * W, H, W, L, ns, W] for i in range(n)], dtype=torch.int64, device=dev)
    cen = torch.tensor([[base[i % 16][1], base[i % 16][2]] for i in range(n)], dtype=torch.float64, device=dev)
    rot = labels24._rot_table(dev)
    pts = torch.empty(n, 24, 2, dtype=torch.int32, device=dev)
    rad = torch.empty(n, 24, dtype=torch.float64, device=dev)
    area = torch.empty(n, dtype=torch.float64, device=dev)

    def run():
        call("ray24", ptr(masks), ptr(desc), ptr(cen), ptr(rot), n, ptr(pts), ptr(rad), stream_ptr())
        call("hull_area24", ptr(pts), n, ptr(area), stream_ptr())
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / 10
    samples = 24.0 * ns * n
    print("GPU: %d objects of %dx%d, %d samples/ray: %.3f ms  = %.0f objects/s, %.1f G samples/s (mask bytes touched %.1f GB/s)"
          % (n, H, W, ns, dt * 1e3, n / dt, samples / dt / 1e9, samples / dt / 1e9))
    from oracle import labels24 as olab
    t0 = time.time()
    k = 4
    for i in range(k):
        wp, wr = olab.rotation_for_24p(base[i][1], base[i][2], base[i][0])
        olab.hull_area(wp)
        assert np.array_equal(pts[i].cpu().numpy(), wp) and np.array_equal(rad[i].cpu().numpy(), wr)
    dc = (time.time() - t0) / k
    print("CPU oracle (coordinate-list form, 1 thread): %.1f ms per object = %.1f objects/s; the GPU/CPU ratio is %.0fx" % (dc * 1e3, 1 / dc, n / dt * dc))


main()
