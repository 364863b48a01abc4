set -e
R=$PWD
OUT=$R/gpurun_out/r3s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EP24_LIB=$R/exploration-of-potential_amd/ep24/libep24_prev.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prev -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prev.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/new -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/new.log 2>&1
for v in prev new; do echo "== $v"; grep "^{" $OUT/$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1); head -12 $f | cut -d, -f1-5 | cut -c1-150; rm -f $(find $OUT/$v -name "*kernel_trace.csv"); done
