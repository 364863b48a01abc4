set -e
R=$PWD
OUT=$R/gpurun_out/r3c
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off $R/tools/hazard_probe.hip -o /tmp/hazard_probe 2> $OUT/hp_build.log
timeout -k 10 150 /tmp/hazard_probe 3000 > $OUT/hazard_v3.txt 2>&1
echo "probe done"
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1 || { tail -30 $OUT/gpu_tests.txt; exit 1; }
tail -3 $OUT/gpu_tests.txt
make -C $R/exploration-of-potential_amd/csrc vec > $OUT/vec_build.log 2>&1
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/novec_$i.json 2> $OUT/novec_$i.err
EP24_LIB=$R/exploration-of-potential_amd/ep24/libep24_vec.so timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/vec_$i.json 2> $OUT/vec_$i.err
done
echo "bench done"
