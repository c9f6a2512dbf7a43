#!/bin/bash
# round 4, GPU session 23 (experiment): what about the per-step read of the resident word costs a stream of solves its throughput
export TMPDIR=/tmp
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule throughput --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; }
for rep in 1 2; do
  echo "off: $(BIOIK_SOLVE_DRAIN_BELOW=0 run 10 60)"
  echo "never triggers, every step: $(BIOIK_SOLVE_DRAIN_BELOW=1 run 10 60)"
  echo "never triggers, one workgroup in eight reads: $(BIOIK_SOLVE_DRAIN_BELOW=1 BIOIK_SOLVE_DRAIN_MIN_STEPS=-1 run 10 60)"
  echo "never triggers, reads another (quiet) word: $(BIOIK_SOLVE_DRAIN_BELOW=1 BIOIK_SOLVE_DRAIN_MIN_STEPS=-2 run 10 60)"
  echo "never triggers, every fourth step: $(BIOIK_SOLVE_DRAIN_BELOW=1 BIOIK_SOLVE_DRAIN_MIN_STEPS=-3 run 10 60)"
done
