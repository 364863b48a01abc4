mkdir -p gpurun_out/nostats
L=$PWD/exploration-of-potential_amd/ep24
for r in 1 2 3; do
  for lib in before ""; do
    echo "== lib ${lib:-new}" >> gpurun_out/nostats/step_ab.txt
    EP24_LIB=$L/libep24${lib:+_$lib}.so timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/nostats/b.out 2> gpurun_out/nostats/b.err || { tail -5 gpurun_out/nostats/b.err; exit 1; }
    python -c "import sys,json; d=json.loads(open('gpurun_out/nostats/b.out').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])" >> gpurun_out/nostats/step_ab.txt
  done
done
paste - - < gpurun_out/nostats/step_ab.txt
