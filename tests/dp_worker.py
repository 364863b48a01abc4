"""One rank of the data-parallel GPU tests (tests/test_gpu_dp.py starts it, once or under torch.distributed.run).

    dp_worker.py --mode single|gloo|nccl --out FILE [--sim-world W]

Every mode trains the same small YOLOX-24p (width 0.25, 192x192, batch 4) with the captured two-lane step:
  phase A  three steps on the SAME batch on every rank -> the N-rank losses must equal the 1-rank losses bit for bit
           (sum of N identical gradients x 1/N is exact);
  phase B  one step from fresh state with a DIFFERENT shard per rank -> the parameters after the update must equal
           `--mode single --sim-world W`, which computes the W shard gradients one after the other in one process, adds
           them in rank order and applies the same fused update with 1/W.
gloo: ranks share cuda:0 (one-GPU box), the collective is gloo; nccl: a one-rank RCCL group (buckets, communication stream
and the nccl backend's stream semantics next to the captured segments, as on 8 GPUs)."""
import argparse
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B, S, G = 4, 192, 6


def build(dev, reducer):
    from ep24 import loss as eloss, nn as enn, train as etrain
    torch.manual_seed(0)
    model = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.25), enn.YOLOXHead(80, 0.25))
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    model.head.initialize_biases(1e-2)
    model.to(dev)
    lf = eloss.Loss_Function(80)
    ts = etrain.TrainStep(model, lf, lr=0.01, momentum=0.9, batch=B, size=S, reducer=reducer)
    return model, ts


def batch(shard, dev):
    from ep24 import synth
    return (synth.make_images(B, S, seed=11 + shard).to(dev), synth.make_labels(B, G, size=S, seed=500 + shard).to(dev))


def crc(t):
    return zlib.crc32(t.detach().cpu().contiguous().numpy().tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True, choices=["single", "gloo", "nccl"])
    ap.add_argument("--out", required=True)
    ap.add_argument("--sim-world", type=int, default=2)
    ap.add_argument("--bf16-wire", action="store_true", help="gradient buckets travel as bf16 (GradReducer comm_dtype)")
    a = ap.parse_args()
    from ep24 import dp
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rank, world = 0, 1
    if a.mode == "gloo":
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
    elif a.mode == "nccl":
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % int(os.environ.get("EP24_TEST_PORT", 29641)), rank=0,
                                world_size=1, device_id=dev)

    def reducer():
        return dp.GradReducer(bucket_bytes=256 << 10, first_bucket_bytes=64 << 10,
                              comm_dtype=torch.bfloat16 if a.bf16_wire else None) if a.mode != "single" else None

    res = {"mode": a.mode, "rank": rank, "world": world}
    # ---- phase A: identical batches
    model, ts = build(dev, reducer())
    if ts.reducer is not None:
        res["buckets"] = len(ts.reducer.buckets)
        res["cuts"] = ts.reducer.cuts()
        res["bwd_len"] = len(ts.eng.bwd)
    img, lab = batch(0, dev)
    losses = []
    for _ in range(3):
        r = ts.step(img, lab)
        torch.cuda.synchronize()
        losses.append(float(r[0]).hex())
    res["losses"] = losses
    res["crc_a"] = crc(ts.home.flat)
    # ---- phase B: one step, a shard per rank
    del model, ts
    model, ts = build(dev, reducer())
    if a.mode == "single":
        W = a.sim_world
        eng = ts.eng
        keep = [b.clone() for b in model.buffers()] + [ts.state.clone()]
        total = torch.zeros_like(ts.home.gflat)
        for r_ in range(W):
            with torch.no_grad():
                for b, k in zip(list(model.buffers()) + [ts.state], keep):
                    b.copy_(k)
            img, lab = batch(r_, dev)
            eng.images.copy_(img)
            ts.labels.copy_(lab)
            ts._phase_forward()
            ts._phase_backward(0, len(eng.bwd))
            torch.cuda.synchronize()
            total += ts.home.gflat
        ts.home.gflat.copy_(total)
        ts.world = W
        ts._hp_dirty = True
        ts._push_hparams()
        ts._phase_update()
    else:
        img, lab = batch(rank, dev)
        ts.step(img, lab)
    torch.cuda.synchronize()
    res["crc_b"] = crc(ts.home.flat)
    res["sum_b"] = float(ts.home.flat.double().sum())
    with open(a.out if world == 1 else "%s.%d" % (a.out, rank), "w") as fh:
        json.dump(res, fh)
    if a.mode != "single":
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
