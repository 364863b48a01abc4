R=$PWD
OUT=$R/gpurun_out/r5m
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_densenet.py tests/test_gpu_engine.py -m gpu -q -x > $OUT/gpu_tests_a.txt 2>&1
tail -4 $OUT/gpu_tests_a.txt
grep -q "passed" $OUT/gpu_tests_a.txt && ! grep -q "failed" $OUT/gpu_tests_a.txt || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/tools/bn_probe.py > $OUT/bn_probe.txt 2>&1
tail -30 $OUT/bn_probe.txt
for i in 1 2; do
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_$i.json 2> $OUT/bench_$i.err
done
cut -c1-200 $OUT/bench_1.json $OUT/bench_2.json
