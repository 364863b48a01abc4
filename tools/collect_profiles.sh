#!/bin/bash
# GPU box: the evidence kept under profiles/ for one round.  usage: tools/collect_profiles.sh TAG GIT_HEAD   (run through gpurun;
# the box has no .git, so the caller passes the commit the tree was pushed from: every summary is stamped with it).  Everything
# lands in gpurun_out/prof_TAG/; tools/summarize_profile.py and tools/summarize_sq.py turn it into the committed summaries.
set -e
TAG=${1:-r04}
HEAD=${2:-unknown}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
echo "$HEAD" > $OUT/git_head.txt
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline > $OUT/bench_sustained.json 2> $OUT/bench_sustained.err
echo "sustained bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
echo "kernel trace done"
EP24_LAYER_TABLE=$OUT/layer_table.txt python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/layer.log 2>&1
echo "layer table done"
# counters: their own passes (never together with a trace domain other than --kernel-trace), eager launches, the same launch lists
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/pmc_f.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/pmc_w.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/sq.log 2>&1
echo "sq pass done"
python3 $R/tools/trace_gaps.py $OUT/kt/kt_kernel_trace.csv > $OUT/stream_gaps.txt 2>&1 || true
python3 $R/tools/trace_step.py $OUT/kt/kt_kernel_trace.csv $OUT/step_timeline.csv > /dev/null 2>&1 || true
# the trace CSVs are large: keep the stats and the counter files only
find $OUT -name "*_kernel_trace.csv" -delete
ls -la $OUT $OUT/kt
