#!/bin/bash
# round 3, GPU session 59: BASELINE.json configs[4] (mixed PR2 / snake batch, sorted by model, end to end through solve_mixed) on one GPU:
# the round-2 size (16384) and the full 262144
O=gpurun_out/s59; mkdir -p $O
export TMPDIR=/tmp
BIOIK_BENCH_C5_BATCH=16384 timeout 600 python bench.py --config c5 --steps 5 --warmup 2 2>/dev/null | grep '^{' | tail -1 > $O/bench_c5_16384.json
timeout 900 python bench.py --config c5 --steps 3 --warmup 1 2>/dev/null | grep '^{' | tail -1 > $O/bench_c5_262144.json
for f in $O/bench_c5_16384.json $O/bench_c5_262144.json; do python -c "
import json; d=json.load(open('$f')); print('$f: %.0f solves/s, %.1f ms per batch' % (d['value'], d['ms_per_step']), d['config'])"; done
