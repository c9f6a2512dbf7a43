#!/bin/bash
# round 5, GPU session 22: the steps a unit runs before it may leave an emptying chip (BIOIK_SOLVE_DRAIN_MIN_STEPS, default 4)
mkdir -p gpurun_out; export TMPDIR=/tmp
for ms in 1 2 4 8; do
  echo -n "min steps $ms: isolated "; BIOIK_SOLVE_DRAIN_MIN_STEPS=$ms python bench.py --timed-only --in-flight 1 --schedule latency --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.3f ms' % (d['value'], d['ms_per_step']), end=' | ')"
  echo -n "three in flight "; BIOIK_SOLVE_DRAIN_MIN_STEPS=$ms python bench.py --timed-only --in-flight 3 --schedule latency --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.3f ms' % (d['value'], d['ms_per_step']), end=' | ')"
  echo -n "driver command "; BIOIK_SOLVE_DRAIN_MIN_STEPS=$ms python bench.py --timed-only --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.3f ms' % (d['value'], d['ms_per_step']))"
done 2>&1 | tee gpurun_out/r05s22_drain_min_steps.log
