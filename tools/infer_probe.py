"""GPU box helper: inference throughput of the N3 row - eval-mode YOLOX-l-24p forward (captured graph) + postprocess."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch
from ep24 import nn as enn, synth
from ep24.infer import postprocess

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
torch.manual_seed(0)
m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
m.head.initialize_biases(1e-2)
m.to(dev).eval()
eng = m.engine(B, S)
eng.images.copy_(synth.make_images(B, S, seed=1).to(dev))
with torch.no_grad():
    eng.forward_eval()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.forward_eval()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / n
    out = eng.outputs.clone()
    # random-init scores are ~0.01: lower the threshold so that NMS has a few hundred candidates per image
    thr = float((out[..., 26] * out[..., 27:].max(-1).values).flatten().kthvalue(int(out.shape[0] * out.shape[1] * 0.97)).values)
    dets = postprocess(out, 80, conf_thre=thr, nms_thre=0.45)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        dets = postprocess(out, 80, conf_thre=thr, nms_thre=0.45)
    torch.cuda.synchronize()
    post_ms = (time.perf_counter() - t0) * 1e3 / n
kept = sum(0 if d is None else d.shape[0] for d in dets)
flops = 2 * 77.694e9 * (S / 640.0) ** 2 * B
print("YOLOX-l-24p eval, B=%d, %dx%d: forward %.2f ms (%.0f img/s, %.0f TFLOP/s), postprocess %.2f ms (%d candidates/img -> %d kept/img), "
      "end to end %.0f img/s" % (B, S, S, fwd_ms, B / fwd_ms * 1e3, flops / fwd_ms / 1e9, post_ms,
                                 int(0.03 * out.shape[1]), kept // B, B / (fwd_ms + post_ms) * 1e3))
