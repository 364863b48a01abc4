set -e
R=$PWD
OUT=$R/gpurun_out/r3a
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off $R/tools/hazard_probe.hip -o /tmp/hazard_probe 2> $OUT/hp_build.log
timeout -k 10 120 /tmp/hazard_probe 3000 > $OUT/hazard.txt 2>&1
echo "probe done"
timeout -k 10 400 python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline > $OUT/sustained.json 2> $OUT/sustained.err
echo "sustained done"
timeout -k 10 400 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
