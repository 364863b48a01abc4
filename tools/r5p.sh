R=$PWD
cd $R
mkdir -p gpurun_out
bash tools/collect_profiles.sh r03 > gpurun_out/prof_r03_collect.log 2>&1
tail -5 gpurun_out/prof_r03_collect.log
cut -c1-300 gpurun_out/prof_r03/bench.json
cut -c1-300 gpurun_out/prof_r03/bench_sustained.json
