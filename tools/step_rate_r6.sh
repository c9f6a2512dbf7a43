#!/bin/bash
# round 6: fixed work (no query may succeed: every unit runs 32 steps) under the throughput schedule's kernels, one launch at a time, at batch sizes that fill the
# chip once / twice / four times: steps per ms chip-wide.  usage: tools/step_rate_r6.sh name=lib[,ENV=v...] ...
for spec in "$@"; do
  n=${spec%%=*}; rest=${spec#*=}; lib=${rest%%,*}; envs=""
  if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
  for b in 4096 8192 16384; do
    v=$(env $envs BIOIK_BENCH_SCHEDULE=throughput BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=$b python bench.py --no-cpu-baseline --timed-only --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms -> %.0f steps/ms chip-wide' % (d['ms_per_step'], $b*32/d['ms_per_step']))")
    echo "$n batch=$b : $v"
  done
done
