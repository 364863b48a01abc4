#!/bin/bash
# GPU box: rocprofv3 kernel stats of the default bench; prints the per-step time of the kernels matching $2 (regex)
set -e
R=$PWD
OUT=$R/${1:-gpurun_out/prof_one}
PAT=${2:-cost_kernel}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
grep "^{" $OUT/kt.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['loss'])"
python3 - <<PY
import csv,re
for r in csv.DictReader(open('$OUT/kt/kt_kernel_stats.csv')):
    if re.search(r'$PAT', r['Name']):
        print("%-60s calls %5s avg %8.1f us  per step %.3f ms" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/15e6))
PY
rm -f $OUT/kt/kt_kernel_trace.csv
