set -e
R=$PWD
OUT=$R/gpurun_out/r3b
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off $R/tools/hazard_probe.hip -o /tmp/hazard_probe 2> $OUT/hp_build.log
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fno-vectorize $R/tools/hazard_probe.hip -o /tmp/hazard_probe_nv 2>> $OUT/hp_build.log
timeout -k 10 120 /tmp/hazard_probe 3000 > $OUT/hazard_v2.txt 2>&1
timeout -k 10 120 /tmp/hazard_probe_nv 3000 > $OUT/hazard_v2_novec.txt 2>&1
echo "probe done"
make -C $R/exploration-of-potential_amd/csrc noload > $OUT/noload_build.log 2>&1
echo "noload built"
cd /tmp && export TMPDIR=/tmp
export EP24_LIB=$R/exploration-of-potential_amd/ep24/libep24_noload.so
timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/noload_off.json 2> $OUT/noload_off.err
EP24_WGRAD_NOLOAD=1 timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/noload_on.json 2> $OUT/noload_on.err
echo "bench done"
EP24_WGRAD_NOLOAD=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_step.py $f $OUT/noload_step.csv
python3 $R/tools/trace_gaps.py $f > $OUT/noload_gaps.txt
rm -f $f
echo "trace done"
