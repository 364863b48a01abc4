mkdir -p gpurun_out/wreg
o=gpurun_out/wreg/configs.txt; rm -f $o
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" > gpurun_out/wreg/c.out 2> gpurun_out/wreg/c.err || { echo "$name FAILED" >> $o; tail -3 gpurun_out/wreg/c.err >> $o; return; }; python -c "import json; d=json.loads(open('gpurun_out/wreg/c.out').read().strip().splitlines()[-1]); print('%-13s %8.2f %7.3f   %s' % ('$name', d['value'], d['ms_per_step'], '$*'))" >> $o; }
run default
run vgg --backbone vgg
run vgg_tiled --backbone vgg --plan conv_kernel_opts=512
run resnet --backbone resnet
run densenet --backbone densenet
run config5 --batch 8 --size 1280 --gts 50 --fisheye
run config5_tiled --batch 8 --size 1280 --gts 50 --fisheye --plan conv_kernel_opts=512
run longrun --long-run
run depthwise_s --depthwise --width 0.5 --depth 0.33
run dense_s --width 0.5 --depth 0.33
cat $o
