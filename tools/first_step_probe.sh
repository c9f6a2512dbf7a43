#!/bin/bash
# cost of the first step() of a solve against the later ones: fixed-work probe (3072 workgroups, no query may succeed) with budgets of 1 ... 16 steps, one launch
for S in 1 2 3 4 8 16; do
BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=$S BIOIK_BENCH_BATCH=3072 python bench.py --no-cpu-baseline --timed-only --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('budget $S steps: %.3f ms per launch' % d['ms_per_step'])"
done
