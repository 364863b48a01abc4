"""GPU box helper (de-risk the 8-GPU configuration on one card): ms per step of the captured YOLOX-l step while a
third stream keeps N whole CUs busy - the footprint RCCL's all-reduce kernels have for most of backward.  Every conv dispatch
heuristic assumes 256 free CUs (one-round grids of 250 halo-patch workgroups, weight-gradient splits that fill the resident slots
exactly once); this measures what happens when they are not.  The hog (tools/cu_hog.hip, built here with hipcc) holds one
1024-thread / 160 KB-LDS workgroup per CU for a fixed time.
usage: cu_contention.py [--steps 10] [--cus 0,4,8,16,32]"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ep24 import loss as eloss, nn as enn, synth, train as etrain  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--cus", default="0,4,8,16,32,0")
ap.add_argument("--light", action="store_true", help="512-thread / 16 KB workgroups (a communication kernel's footprint) instead of whole CUs")
a = ap.parse_args()

so = "/tmp/libcu_hog.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "tools", "cu_hog.hip"), "-o", so], check=True)
hog = ctypes.CDLL(so).cu_hog_light if a.light else ctypes.CDLL(so).cu_hog
hog.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
hog.restype = ctypes.c_int

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
model.head.initialize_biases(1e-2)
model.to(dev)
ts = etrain.TrainStep(model, eloss.Loss_Function(80), lr=0.001, momentum=0.9, batch=20, size=640)
ts.eng.images.copy_(synth.make_images(20, 640, seed=1).to(dev))
ts.labels.copy_(synth.make_labels(20, 10, size=640, seed=1000).to(dev))
for _ in range(5):
    ts.step()
torch.cuda.synchronize()
sink = torch.zeros(4, dtype=torch.int32, device=dev)
hs = torch.cuda.Stream()
print("# ms per step of the captured YOLOX-l step (B = 20, 640x640) with N %s held by another stream; %d steps per row"
      % ("512-thread / 16 KB workgroups" if a.light else "whole CUs (1024 threads, 160 KB LDS)", a.steps))
print("%8s %10s %10s" % ("busy CUs", "ms/step", "vs 0"))
base = None
for n in [int(v) for v in a.cus.split(",")]:
    torch.cuda.synchronize()
    rc = hog(n, 60.0 * a.steps + 100.0, sink.data_ptr(), hs.cuda_stream)       # outlasts the timed steps, then drains by itself
    assert rc == 0, rc
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        ts.step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    base = base or ms
    print("%8d %10.3f %10.3f" % (n, ms, ms / base), flush=True)
