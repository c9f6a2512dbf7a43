#!/bin/bash
# round 3, GPU session 50 (the same as 37, with the automatic ten solves in flight): dry run of the N = 2 control flow of bench.py on a one-GPU box (two ranks sharing the device, gloo for the barrier and the reductions)
O=gpurun_out/s50; mkdir -p $O
export TMPDIR=/tmp
BIOIK_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_two_ranks.json 2> $O/bench_two_ranks.err
echo rc=$?
python -c "
import json; d=json.loads([l for l in open('$O/bench_two_ranks.json') if l.startswith('{')][-1])
print('two ranks on one GPU (gloo): %.0f solves/s %.2f ms n_gpus %d in flight %d' % (d['value'], d['ms_per_step'], d['n_gpus'], d['config']['batches_in_flight']))"
tail -3 $O/bench_two_ranks.err
