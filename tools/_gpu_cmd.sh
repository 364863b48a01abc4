mkdir -p gpurun_out/tiled
L=$PWD/exploration-of-potential_amd/ep24
timeout -k 10 400 python -m pytest tests/test_gpu_conv.py -x -q > gpurun_out/tiled/tests.log 2>&1; rc=$?; tail -3 gpurun_out/tiled/tests.log; [ $rc = 0 ] || exit $rc
S="20,20,512,512,3,1;20,40,512,256,1,1;20,20,2048,1024,1,1;20,20,1024,512,1,1;20,80,256,512,3,2;20,40,512,1024,3,2;20,160,128,256,3,2"
for r in 1 2; do
for lib in before ""; do
  echo "== lib ${lib:-new}" >> gpurun_out/tiled/ab.txt
  EP24_AB_KOS=0 EP24_LIB=$L/libep24${lib:+_$lib}.so timeout -k 10 200 python tools/wreg_ab.py "$S" >> gpurun_out/tiled/ab.txt 2>&1
done; done
grep -v "amdgpu.ids\|^kind\|ring timeouts" gpurun_out/tiled/ab.txt
for r in 1 2 3; do
  for lib in before ""; do
    echo "== lib ${lib:-new}" >> gpurun_out/tiled/step_ab.txt
    EP24_LIB=$L/libep24${lib:+_$lib}.so timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/tiled/b.out 2> gpurun_out/tiled/b.err || { tail -5 gpurun_out/tiled/b.err; exit 1; }
    python -c "import sys,json; d=json.loads(open('gpurun_out/tiled/b.out').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['members']['igemm_dma_kernel']['achieved'], d['loss'])" >> gpurun_out/tiled/step_ab.txt
  done
done
cat gpurun_out/tiled/step_ab.txt
