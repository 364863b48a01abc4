"""GPU box helper (round 5): does the trainer's input path hold when EIGHT ranks share one host?

One real trainer (train_24p.py, BASELINE config 2, the default raw uint8 source + GPU letterbox behind the side-stream prefetcher)
feeds the GPU; N - 1 "sink" processes run the SAME host side of that path - the Exp's loader, the collate, and a copy of every
image into a staging buffer (what the driver's pageable upload does on the host) - without touching the GPU (a one-GPU box lets few
processes use its card), UNPACED: their rate is the headroom a rank's host side has while 7 others do the same.  The box gives a
one-GPU call 16 host cores; an 8-GPU node has 8 x that.

usage: trainer_8sets.py [--sets 8] [--steps 300] [--out profiles/r05_trainer_8sets.json]
       trainer_8sets.py --sink SECONDS          (internal)"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Y24 = os.path.join(ROOT, "exploration-of-potential_amd", "yolox_24p")
EXP = os.path.join(Y24, "load_train", "yolox_24p_l_train.py")


def sink(seconds, batch=20, fp32=False):
    sys.path.insert(0, Y24)
    import torch
    from exp import get_exp
    torch.set_num_threads(1)                                  # a rank's training thread is one thread
    exp = get_exp(EXP)
    exp.synthetic_len = 64 * batch
    loader = exp.get_data_loader(batch, raw_u8=not fp32, workers=0 if not fp32 else None)
    S = exp.input_size
    stage = torch.empty(batch, S[0], S[1], 3, dtype=torch.uint8) if not fp32 else torch.empty(batch, 3, S[0], S[1])
    n, t0 = 0, time.perf_counter()
    t_first = None
    while True:
        for images, labels, _info, _ids in loader:
            if fp32:
                stage.copy_(images)
            else:
                for i, im in enumerate(images):
                    stage[i].copy_(im)
            if t_first is None:                               # the first pass builds the 64 cached items
                t_first = time.perf_counter()
                n = 0
            n += batch
            if time.perf_counter() - t0 > seconds:
                dt = time.perf_counter() - t_first
                print(json.dumps({"images": n, "seconds": round(dt, 3), "images_per_s": round(n / dt, 1)}))
                return


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sink", type=float, default=0.0)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trainer_8sets.json"))
    a = ap.parse_args()
    if a.sink:
        return sink(a.sink, fp32=a.fp32)
    res = {}
    for label, n_sinks in (("alone", 0), ("with_%d_sinks" % (a.sets - 1), a.sets - 1)):
        tp = os.path.join("/tmp", "tp_%s.json" % label)
        secs = 25 + a.steps * 0.03
        sinks = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--sink", str(secs)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for _ in range(n_sinks)]
        env = dict(os.environ)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.join(Y24, "train_24p.py"), "-f", EXP, "-b", "20", "-l", "0.01", "--synthetic", "--steps", str(a.steps),
                            "--log-interval", "100", "--throughput-json", tp, "--throughput-window", str(a.steps - 50), "--output-dir", "/tmp/ep24_8sets",
                            "--synthetic-len", str(20 * (a.steps + 10))], cwd=Y24, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        wall = time.time() - t0
        assert p.returncode == 0, p.stdout[-3000:]
        rec = {"trainer": json.load(open(tp)), "trainer_wall_s": round(wall, 1), "sinks": []}
        for s in sinks:
            out, _ = s.communicate(timeout=300)
            lines = [ln for ln in out.splitlines() if ln.startswith("{")]
            rec["sinks"].append(json.loads(lines[-1]) if lines else {"error": out[-300:]})
        res[label] = rec
        print(label, "trainer %.1f images/s (%.2f ms per step; host loader+upload / enqueue / rest %s)" % (
            rec["trainer"]["images_per_s"], rec["trainer"]["ms_per_step"], rec["trainer"]["host_ms_per_step"]),
            "sinks:", [s.get("images_per_s") for s in rec["sinks"]])
    res["host_cpus"] = os.cpu_count()
    res["note"] = ("one GPU-fed trainer (train_24p.py defaults: raw uint8 source, GPU letterbox, side-stream prefetch) + N - 1 unpaced host-side sinks "
                   "(loader + collate + staging copy of every image, one thread each)")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
