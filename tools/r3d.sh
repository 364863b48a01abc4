R=$PWD
OUT=$R/gpurun_out/r3d
mkdir -p $OUT
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1
tail -5 $OUT/gpu_tests.txt
timeout -k 10 600 python3 tools/ring_ab.py > $OUT/ring_ab.txt 2>&1
tail -5 $OUT/ring_ab.txt
make -C $R/exploration-of-potential_amd/csrc vec > $OUT/vec_build.log 2>&1
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/novec_$i.json 2> $OUT/novec_$i.err
EP24_LIB=$R/exploration-of-potential_amd/ep24/libep24_vec.so timeout -k 10 300 python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline > $OUT/vec_$i.json 2> $OUT/vec_$i.err
done
echo "bench done"
