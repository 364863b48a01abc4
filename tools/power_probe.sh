#!/bin/bash
# GPU box: sample board power and shader clock while bench.py runs (is the step held back by the power cap?)
OUT=${1:-gpurun_out/power}
STEPS=${2:-2500}
mkdir -p $OUT
python bench.py --no-cpu-baseline --steps $STEPS > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
: > $OUT/smi.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Package Power\|sclk\|mclk" | sed 's/^GPU\[0\][ \t]*: //' | tr '\n' ';' >> $OUT/smi.txt
  echo >> $OUT/smi.txt
  sleep 0.7
done
wait $BP
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" >> $OUT/smi.txt
cat $OUT/smi.txt
python -c "import json;d=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][0]);print(d['value'],d['ms_per_step'])"
