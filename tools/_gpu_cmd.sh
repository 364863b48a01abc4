mkdir -p gpurun_out/s33
L=$PWD/exploration-of-potential_amd/ep24
timeout -k 10 500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_engine.py -x -q > gpurun_out/s33/tests.log 2>&1; rc=$?; tail -2 gpurun_out/s33/tests.log; [ $rc = 0 ] || exit $rc
for r in 1 2 3; do
  for lib in before ""; do
    echo "== lib ${lib:-new}" >> gpurun_out/s33/step_ab.txt
    EP24_LIB=$L/libep24${lib:+_$lib}.so timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s33/b.out 2> gpurun_out/s33/b.err || { tail -5 gpurun_out/s33/b.err; exit 1; }
    python -c "import sys,json; d=json.loads(open('gpurun_out/s33/b.out').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> gpurun_out/s33/step_ab.txt
  done
done
paste - - < gpurun_out/s33/step_ab.txt
R=$PWD
for lib in before ""; do
  EP24_LIB=$L/libep24${lib:+_$lib}.so bash tools/prof_one.sh gpurun_out/s33/prof_${lib:-new} 'wgrad_kernel<64, (64|192), true>|wgrad_reduce' || exit 1
  cd $R
done
