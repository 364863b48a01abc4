#!/bin/bash
# Rehearsal of the data-parallel step on a ONE-GPU box: N ranks share cuda:0, the collective is gloo, every rank gets
# the same batch, so the N-rank loss must equal the 1-rank loss bit for bit (sum of N identical gradients x 1/N).
# Exercises everything of the multi-GPU path except RCCL itself: bucket planning, backward segments as graphs, the
# all-reduce issued from the weight-gradient lane, the 1/world scale inside the fused SGD.
# usage: tools/dp_rehearsal.sh [ranks=2] [steps=4]
set -e
N=${1:-2}; K=${2:-4}
cd "$(dirname "$0")/.."
python bench.py --steps $K --warmup 2 --no-cpu-baseline | tail -1 | python -c "import json,sys; print('1 rank :', json.loads(sys.stdin.read())['loss'])"
EP24_REHEARSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus $N --steps $K --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; print('$N ranks:', json.loads(sys.stdin.read())['loss'])"
