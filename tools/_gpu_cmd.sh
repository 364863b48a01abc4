mkdir -p gpurun_out/s38
L=$PWD/exploration-of-potential_amd/ep24
timeout -k 10 500 python -m pytest tests/test_gpu_loss.py tests/test_gpu_engine.py tests/test_gpu_fp32.py tests/test_gpu_hazard.py tests/test_gpu_fullsize.py -x -q > gpurun_out/s38/tests.log 2>&1; rc=$?; tail -2 gpurun_out/s38/tests.log; [ $rc = 0 ] || exit $rc
for lib in before ""; do
  echo "== lib ${lib:-new}"
  EP24_LIB=$L/libep24${lib:+_$lib}.so timeout -k 10 200 python tools/loss_time.py > gpurun_out/s38/loss_time_${lib:-new}.txt 2>&1 || { tail -5 gpurun_out/s38/loss_time_${lib:-new}.txt; exit 1; }
  grep -i "dynamic_k\|sum" gpurun_out/s38/loss_time_${lib:-new}.txt
done
for r in 1 2 3; do
  for lib in before ""; do
    echo "== lib ${lib:-new}" >> gpurun_out/s38/step_ab.txt
    EP24_LIB=$L/libep24${lib:+_$lib}.so timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s38/b.out 2> gpurun_out/s38/b.err || { tail -5 gpurun_out/s38/b.err; exit 1; }
    python -c "import sys,json; d=json.loads(open('gpurun_out/s38/b.out').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['loss'])" >> gpurun_out/s38/step_ab.txt
  done
done
paste - - < gpurun_out/s38/step_ab.txt
