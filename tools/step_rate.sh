#!/bin/bash
# kernel speed independent of how the random search happens to go: no query may succeed (dtwist 1e-300), every workgroup runs
# exactly 32 steps; chip-wide steps/ms at 3072 workgroups and the time per step of one lone workgroup.  usage: tools/step_rate.sh lib...
# (the libraries are visited ROUNDS times in turn, 20 timed launches each: single short runs differ by up to 5 % on one box)
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do for lib in "$@"; do for b in 1 3072; do
  v=$(BIOIK_BENCH_SCHEDULE=${SCHEDULE:-latency} BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=$b python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms -> %.1f us per step per workgroup, %.0f steps/ms chip-wide' % (d['ms_per_step'], d['ms_per_step']*1e3/32, $b*32/d['ms_per_step']))")
  echo "$lib batch=$b : $v"
done; done; done
