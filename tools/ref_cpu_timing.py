"""the REFERENCE's own training step on the build container's CPU cores - YOLOX-l-24p, B = 1, 640x640,
fp32, 5 synthetic ground truths; 1 warm-up + N timed steps, median.  The reference source is imported unmodified from
/root/reference through the harness of tests/golden/make_golden.py; the step body is
yolox_24p/train_24p.py:86-104 (zero_grad, model(images, train=True), Loss_Function.forward, backward, SGD nesterov step).
Build container only (the reference never travels to the GPU box); writes profiles/<tag>_reference_cpu_timing.json.
usage: PYTHONDONTWRITEBYTECODE=1 python tools/ref_cpu_timing.py [tag=r02] [steps=5]"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
import make_golden  # noqa: E402
from ep24 import synth  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    torch.set_num_threads(os.cpu_count())
    utils, models = make_golden.load_reference()
    torch.manual_seed(0)
    in_ch = [256, 512, 1024]
    model = models.YOLOX(models.YOLOPAFPN(1.0, 1.0, in_channels=in_ch), models.YOLOXHead(80, 1.0, in_channels=in_ch))
    for m in model.modules():                                      # Exp.get_model: init_yolo + bias prior (exp/yolox_base.py:55-72)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    model.head.initialize_biases(1e-2)
    model.train()
    loss_fn = models.Loss_Function(80)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, nesterov=True)       # exp/yolox_base.py:120-124
    images = synth.make_images(1, 640, seed=1)
    labels = synth.make_labels(1, 5, seed=2)
    rec = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        out = model(images, train=True)
        t1 = time.perf_counter()
        tup = loss_fn.forward(out, labels)
        t2 = time.perf_counter()
        tup[0].backward()
        t3 = time.perf_counter()
        opt.step()
        t4 = time.perf_counter()
        if i:
            rec.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, float(tup[0])))
        print("step %d: fwd %.2f s, loss %.2f s, bwd %.2f s, sgd %.2f s, total %.2f s, loss %.4f" %
              (i, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, float(tup[0])), flush=True)
    med = [statistics.median(r[k] for r in rec) for k in range(5)]
    res = {"what": "reference train_24p.py step body on CPU (reference source, unmodified, imported through the oracle harness)",
           "model": "YOLOX-l-24p (depth 1.0, width 1.0), 54.2 M parameters", "batch": 1, "size": 640, "gts": 5, "dtype": "fp32",
           "torch": torch.__version__, "threads": torch.get_num_threads(), "host_cores": os.cpu_count(), "warmup_steps": 1,
           "timed_steps": steps, "median_s": {"forward": round(med[0], 3), "loss": round(med[1], 3), "backward": round(med[2], 3),
                                              "sgd": round(med[3], 3), "step": round(med[4], 3)},
           "images_per_s": round(1.0 / med[4], 4), "losses": [round(r[5], 4) for r in rec]}
    path = os.path.join(ROOT, "profiles", "%s_reference_cpu_timing.json" % tag)
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res))


main()
