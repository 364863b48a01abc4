# final-tree record: the other configurations and the entry point's default mode beside bench.py, one box
set -e
R=$PWD; O=$R/gpurun_out/s35; mkdir -p $O
cfg() { n=$1; shift; timeout -k 10 240 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" > $O/c_$n.json 2> $O/c_$n.err || { tail -5 $O/c_$n.err; exit 1; }
  python -c "import json,sys; d=json.loads([l for l in open('$O/c_$n.json') if l.startswith('{')][-1]); print('%-14s %8.2f %7.3f   %s' % ('$n', d['value'], d['ms_per_step'], ' '.join(sys.argv[1:])))" "$@" | tee -a $O/configs.txt; }
cfg default
cfg vgg --backbone vgg
cfg resnet --backbone resnet
cfg densenet --backbone densenet
cfg config5 --batch 8 --size 1280 --gts 50 --fisheye
cfg longrun --long-run
cd $R/exploration-of-potential_amd/yolox_24p
timeout -k 10 300 python3 train_24p.py -f load_train/yolox_24p_l_train.py -b 20 -l 0.01 --synthetic --steps 250 --synthetic-len 6000 --log-interval 50 --output-dir $O/run --throughput-json $O/tp_raw_u8.json > $O/raw_u8.log 2>&1 || { tail -5 $O/raw_u8.log; exit 1; }
rm -rf $O/run
cd $R
python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_sustained.json 2> $O/bench.err
python3 -c "
import json
t=json.load(open('$O/tp_raw_u8.json')); b=json.loads([l for l in open('$O/bench_sustained.json') if l.startswith('{')][0])
print('trainer', t['images_per_s'], 'bench', b['value'], 'ratio', round(t['images_per_s']/b['value'],4))"
