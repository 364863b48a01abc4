R=$PWD
OUT=$R/gpurun_out/r5n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_1.json 2> $OUT/bench_1.err
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --plan chunked_update=0 > $OUT/bench_nochunk.json 2> $OUT/bench_nochunk.err
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_2.json 2> $OUT/bench_2.err
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --plan chunked_update=0 > $OUT/bench_nochunk2.json 2> $OUT/bench_nochunk2.err
cut -c1-200 $OUT/bench_1.json $OUT/bench_nochunk.json $OUT/bench_2.json $OUT/bench_nochunk2.json
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_step.py $f $OUT/step.csv
python3 $R/tools/trace_gaps.py $f > $OUT/gaps.txt; cat $OUT/gaps.txt
rm -f $f
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_n2.py tests/test_gpu_trainer.py tests/test_gpu_dp.py -m gpu -q > $OUT/gpu_tests.txt 2>&1
tail -6 $OUT/gpu_tests.txt
echo "done"
