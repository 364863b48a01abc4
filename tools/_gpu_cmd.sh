mkdir -p gpurun_out/s32
L=$PWD/exploration-of-potential_amd/ep24
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_xf.py tests/test_gpu_engine.py -x -q > gpurun_out/s32/tests.log 2>&1; rc=$?; tail -2 gpurun_out/s32/tests.log; [ $rc = 0 ] || exit $rc
for r in 1 2 3; do
  for lib in before ""; do
    echo "== lib ${lib:-new}" >> gpurun_out/s32/step_ab.txt
    EP24_LIB=$L/libep24${lib:+_$lib}.so timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s32/b.out 2> gpurun_out/s32/b.err || { tail -5 gpurun_out/s32/b.err; exit 1; }
    python -c "import sys,json; d=json.loads(open('gpurun_out/s32/b.out').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['families']['igemm_stream_kernel']['ms_per_step'])" >> gpurun_out/s32/step_ab.txt
  done
done
paste - - < gpurun_out/s32/step_ab.txt
