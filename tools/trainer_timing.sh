#!/bin/bash
# GPU box: images/s of the ENTRY POINT (train_24p.py, BASELINE config 2: YOLOX-l-24p, B = 20, 640 x 640) over the last 200 of 250
# steps, in its three input modes, beside bench.py on the same box.  usage: tools/trainer_timing.sh OUTDIR
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/${1:-gpurun_out/trainer}
mkdir -p $OUT
cd $R/exploration-of-potential_amd/yolox_24p
COMMON="-f load_train/yolox_24p_l_train.py -b 20 -l 0.01 --synthetic --steps 250 --synthetic-len 6000 --log-interval 50 --output-dir $OUT/run"
run() {   # name, flags
  n=$1; shift
  timeout -k 10 300 python3 train_24p.py $COMMON "$@" --throughput-json $OUT/tp_$n.json > $OUT/$n.log 2>&1
  echo "$n done"
}
run raw_u8                                            # the default since round 5: raw uint8 source, GPU letterbox, side-stream upload
run raw_u8_w4 --loader-workers 4
run prefetch --fp32-batches                           # round 4's default: loader processes, page-locked fp32 batches, side-stream upload
run prefetch_w0 --fp32-batches --loader-workers 0
run prefetch_w4 --fp32-batches --loader-workers 4 --loader-pin 0
run prefetch_w4_pin --fp32-batches --loader-workers 4 --loader-pin 1
run no_prefetch --no-prefetch --loader-workers 0
rm -rf $OUT/run
cd $R
python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > $OUT/bench_sustained.json 2> $OUT/bench.err
python3 - <<PY
import json
MODES = ("prefetch", "prefetch_w0", "prefetch_w4", "prefetch_w4_pin", "raw_u8", "raw_u8_w4", "no_prefetch")
o = {"command": "tools/trainer_timing.sh: train_24p.py -f load_train/yolox_24p_l_train.py -b 20 -l 0.01 --synthetic --steps 250 --synthetic-len 6000 --log-interval 50 [mode]; window = the last 200 steps, synchronised at both ends"}
for k in MODES:
    o[k] = json.load(open("$OUT/tp_%s.json" % k))
b = json.loads([l for l in open("$OUT/bench_sustained.json") if l.startswith("{")][0])
o["bench_py_same_box"] = {"images_per_s": b["value"], "ms_per_step": b["ms_per_step"], "steps": b["steps"], "warmup": b["warmup"]}
for k in MODES:
    o[k]["vs_bench"] = round(o[k]["images_per_s"] / b["value"], 4)
json.dump(o, open("$OUT/trainer.json", "w"), indent=1)
print(json.dumps(o, indent=1))
PY
