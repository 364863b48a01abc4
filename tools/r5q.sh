R=$PWD
OUT=$R/gpurun_out/r5q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
cut -c1-260 $OUT/bench.json
python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline > $OUT/bench_sustained.json 2> $OUT/bench_sustained.err
cut -c1-260 $OUT/bench_sustained.json
for bb in resnet densenet vgg; do
python3 $R/bench.py --backbone $bb --no-cpu-baseline > $OUT/bench_$bb.json 2> $OUT/bench_$bb.err
cut -c1-200 $OUT/bench_$bb.json
done
python3 $R/bench.py --batch 8 --size 1280 --gts 50 --fisheye --no-cpu-baseline > $OUT/bench_config5.json 2> $OUT/bench_config5.err
cut -c1-200 $OUT/bench_config5.json
python3 $R/bench.py --long-run --no-cpu-baseline > $OUT/bench_longrun.json 2> $OUT/bench_longrun.err
cut -c1-200 $OUT/bench_longrun.json
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/sq.log 2>&1
f=$(find $OUT/sq -name "*counter_collection.csv" | head -1)
python3 $R/tools/summarize_sq.py $f $OUT/sq.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph"
rm -f $f $(find $OUT/sq -name "*kernel_trace.csv")
echo done
