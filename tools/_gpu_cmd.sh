mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final/full_tests.log 2>&1; rc=$?
tail -3 gpurun_out/final/full_tests.log
[ $rc = 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
bash tools/collect_profiles.sh r05h ba8b03c > gpurun_out/collect_r05h.log 2>&1; tail -2 gpurun_out/collect_r05h.log
python -c "import json; d=json.load(open('gpurun_out/prof_r05h/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
o=gpurun_out/final/configs.txt; rm -f $o
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" > gpurun_out/final/c.out 2> gpurun_out/final/c.err || { echo "$name FAILED" >> $o; tail -3 gpurun_out/final/c.err >> $o; return; }; python -c "import json; d=json.loads(open('gpurun_out/final/c.out').read().strip().splitlines()[-1]); print('%-13s %8.2f %7.3f   %s' % ('$name', d['value'], d['ms_per_step'], '$*'))" >> $o; }
run default
run vgg --backbone vgg
run resnet --backbone resnet
run densenet --backbone densenet
run config5 --batch 8 --size 1280 --gts 50 --fisheye
run longrun --long-run
run depthwise_s --depthwise --width 0.5 --depth 0.33
run dense_s --width 0.5 --depth 0.33
cat $o
