mkdir -p gpurun_out/s39
L=$PWD/exploration-of-potential_amd/ep24
timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -k spp -x -q > gpurun_out/s39/tests_spp.log 2>&1; rc=$?; tail -1 gpurun_out/s39/tests_spp.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python -m pytest tests/test_gpu_engine.py -x -q > gpurun_out/s39/tests.log 2>&1; rc=$?; tail -1 gpurun_out/s39/tests.log; [ $rc = 0 ] || exit $rc
for r in 1 2; do
  for lib in before ""; do
    echo "== lib ${lib:-new}" >> gpurun_out/s39/step_ab.txt
    EP24_LIB=$L/libep24${lib:+_$lib}.so timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s39/b.out 2> gpurun_out/s39/b.err || { tail -5 gpurun_out/s39/b.err; exit 1; }
    python -c "import sys,json; d=json.loads(open('gpurun_out/s39/b.out').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['loss'])" >> gpurun_out/s39/step_ab.txt
  done
done
paste - - < gpurun_out/s39/step_ab.txt
R=$PWD
for lib in before ""; do
  EP24_LIB=$L/libep24${lib:+_$lib}.so bash tools/prof_one.sh gpurun_out/s39/prof_${lib:-new} 'spp_(row|col)_kernel' || exit 1
  cd $R
done
