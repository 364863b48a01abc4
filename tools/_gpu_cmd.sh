mkdir -p gpurun_out/ring
L=$PWD/exploration-of-potential_amd/ep24
EP24_LIB=$L/libep24_stamps.so timeout -k 10 200 python tools/ring_estamps.py > gpurun_out/ring/estamps4.txt 2>&1
grep -v amdgpu.ids gpurun_out/ring/estamps4.txt
timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -x -q -k "hot_shapes or infer_unit" > gpurun_out/ring/tests.log 2>&1; tail -2 gpurun_out/ring/tests.log
for r in 1 2 3; do
  for lib in before ""; do
    echo "== lib ${lib:-new}" >> gpurun_out/ring/step_ab2.txt
    EP24_LIB=$L/libep24${lib:+_$lib}.so timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/ring/b.out 2> gpurun_out/ring/b.err || { tail -5 gpurun_out/ring/b.err; exit 1; }
    python -c "import sys,json; d=json.loads(open('gpurun_out/ring/b.out').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['members']['conv_ring_kernel']['achieved'])" >> gpurun_out/ring/step_ab2.txt
  done
done
paste - - < gpurun_out/ring/step_ab2.txt
