mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final/full_tests.log 2>&1; rc=$?
tail -3 gpurun_out/final/full_tests.log
[ $rc = 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
bash tools/collect_profiles.sh r05g 6518934 > gpurun_out/collect_r05g.log 2>&1; tail -2 gpurun_out/collect_r05g.log
python -c "import json; d=json.load(open('gpurun_out/prof_r05g/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
