R=$PWD
OUT=$R/gpurun_out/r5d
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "fused_bn" > $OUT/gpu_tests_a.txt 2>&1
tail -5 $OUT/gpu_tests_a.txt
grep -q "passed" $OUT/gpu_tests_a.txt && ! grep -q "failed" $OUT/gpu_tests_a.txt || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_1.json 2> $OUT/bench_1.err
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --plan fuse_bn_reduce=0 > $OUT/bench_nofuse.json 2> $OUT/bench_nofuse.err
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_2.json 2> $OUT/bench_2.err
cut -c1-200 $OUT/bench_1.json $OUT/bench_nofuse.json $OUT/bench_2.json
cd $R
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > $OUT/gpu_tests_b.txt 2>&1
tail -5 $OUT/gpu_tests_b.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_step.py $f $OUT/step.csv
python3 $R/tools/trace_gaps.py $f > $OUT/gaps.txt; cat $OUT/gaps.txt
rm -f $f
echo "done"
