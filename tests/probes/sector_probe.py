"""GPU box helper: per-image time of the sector warp (a13) on device-resident inputs vs the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ep24 import sector as esec

dev = torch.device("cuda", 0)
D = esec.Image_Distortion("cuda:0")
for (h, w, theta) in [(1280, 1280, 60), (640, 640, 90), (1280, 1280, 180)]:
    rng = np.random.RandomState(0)
    img = torch.from_numpy(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).to(dev)
    msk = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
    msk[h // 4: h // 2, w // 4: w // 2] = 255
    winner, cw, box, T = D._map(theta, h, w, None)
    for _ in range(3):
        a = D._warp(img, winner, cw, box, T, 114); b = D._warp(msk, winner, cw, box, T, 0)
    torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        a = D._warp(img, winner, cw, box, T, 114); b = D._warp(msk, winner, cw, box, T, 0)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    oh, ow = a.shape[0], a.shape[1]
    alg = 2 * (h * w * 3 + oh * ow * 3) + oh * ow * 4            # image+mask in, outputs out, winner map
    moved = 2 * (h * w * 3 + 2 * T * esec.N_ANG * 3 + oh * ow * 3) + oh * ow * 4
    line = "%dx%d theta %3d -> %dx%d (T=%d): GPU %.1f us/img (image+mask); algorithmic %.1f MB = %.0f GB/s, moved incl. resized intermediate %.1f MB = %.0f GB/s" % (
        h, w, theta, oh, ow, T, us, alg / 1e6, alg / us / 1e3, moved / 1e6, moved / us / 1e3)
    if os.environ.get("CPU"):
        from oracle import sector as osec
        t0 = time.perf_counter()
        osec.sector_distort(img.cpu().numpy(), msk.cpu().numpy(), theta)
        line += "; CPU oracle %.2f s" % (time.perf_counter() - t0)
    print(line, flush=True)
