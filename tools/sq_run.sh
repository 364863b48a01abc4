set -e
R=$PWD
OUT=$R/gpurun_out/r4x
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/sq.log 2>&1
f=$(find $OUT/sq -name "*counter_collection.csv" | head -1)
python3 $R/tools/summarize_sq.py $f $OUT/sq.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph"
rm -f $f $(find $OUT/sq -name "*kernel_trace.csv")
