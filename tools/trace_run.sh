set -e
R=$PWD
OUT=$R/gpurun_out/r3x
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_step.py $f $OUT/step.csv
python3 $R/tools/trace_gaps.py $f > $OUT/gaps.txt; cat $OUT/gaps.txt
rm -f $f
