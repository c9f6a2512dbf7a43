#!/bin/bash
# round 4, GPU session 21: what the hand-over machinery costs when it never triggers (BIOIK_SOLVE_DRAIN_BELOW=1), steady state
O=gpurun_out/r04s21; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule throughput --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; }
for rep in 1 2; do for n in 0 1 256; do
  export BIOIK_SOLVE_DRAIN_BELOW=$n
  echo "drain below $n: 60 steps, 10 in flight $(run 10 60) | 60 steps, 3 in flight $(run 3 60)"
done; done 2>&1 | tee -a $O/drain_cost.log
