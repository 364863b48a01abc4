mkdir -p gpurun_out/wreg
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/wreg/full_tests.log 2>&1; rc=$?
tail -5 gpurun_out/wreg/full_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/wreg/bench_default.json 2> gpurun_out/wreg/bench_default.err || { tail -5 gpurun_out/wreg/bench_default.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/wreg/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['loss'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['frac_replayed'], d['cpu_baseline']['value'])"
