"""GPU box helper: one training step of a YOLOX network on fixed synthetic inputs with the library named by EP24_LIB; writes the head outputs,
the loss and the flat gradient to a file, or compares two such files (relative rms / max differences).
usage: numeric_ab.py run OUT.pt [width depth size batch] | numeric_ab.py cmp A.pt B.pt
(What it can say: a library is reproducible bit for bit from run to run.  What it cannot: at random initialisation this network amplifies
one-ulp differences of a BatchNorm statistic into O(0.2) relative differences at the head and into other SimOTA assignments - the bridge
of profiles/r04_bridge.txt - so two libraries that differ in a sum's rounding read "rel rms 0.19, gradient cosine 0.03" here.  Parity is
what the teacher-forced unit tests against the oracle check.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402


def run(out, width=1.0, depth=1.0, size=640, batch=4):
    from ep24 import loss as eloss, nn as enn, synth, train as etrain
    dev = "cuda:0"
    torch.manual_seed(0)
    model = enn.YOLOX(enn.YOLOPAFPN(depth, width), enn.YOLOXHead(80, width))
    model.head.initialize_biases(1e-2)
    model.to(dev)
    images = synth.make_images(batch, size, seed=1).to(dev)
    labels = synth.make_labels(batch, [4, 2, 6, 3][:batch] if batch <= 4 else [5] * batch, size=size, seed=2)
    lf = eloss.Loss_Function(80)
    step = etrain.TrainStep(model, lf, lr=0.0, momentum=0.0, batch=batch, size=size)
    res = step.step(images, labels.to(dev))
    torch.cuda.synchronize()
    torch.save({"out": step.eng.outputs.detach().float().cpu(), "loss": float(res[0]), "grad": step.home.gflat.detach().float().cpu() if hasattr(step.home, "gflat") else None}, out)
    print("loss", float(res[0]))


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    for k in ("out", "grad"):
        if A[k] is None or B[k] is None:
            continue
        x, y = A[k].double(), B[k].double()
        d = (x - y)
        print("%-5s rel rms %.3e   max abs diff %.3e   max abs %.3e   cos %.8f" % (k, float(d.pow(2).mean().sqrt() / y.pow(2).mean().sqrt()), float(d.abs().max()), float(y.abs().max()),
                                                                          float((x * y).sum() / (x.norm() * y.norm()))))
    print("loss %.6f %.6f  rel %.3e" % (A["loss"], B["loss"], abs(A["loss"] - B["loss"]) / abs(B["loss"])))


if sys.argv[1] == "run":
    run(sys.argv[2], *[float(v) if i < 2 else int(v) for i, v in enumerate(sys.argv[3:])])
else:
    cmp(sys.argv[2], sys.argv[3])
