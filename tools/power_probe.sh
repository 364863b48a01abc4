#!/bin/bash
# GPU box: sample board power and shader clock while bench.py runs (is the step held back by the power cap?)
OUT=${1:-gpurun_out/power}
mkdir -p $OUT
python bench.py --no-cpu-baseline --steps 400 > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
sleep 25
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -i "power\|sclk\|Temperature (Sensor junction)\|cap" | tr '\n' ';' >> $OUT/smi.txt
  echo >> $OUT/smi.txt
  sleep 1
done
wait $BP
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" >> $OUT/smi.txt
cat $OUT/smi.txt
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'])"
