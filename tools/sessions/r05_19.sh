#!/bin/bash
# round 5, GPU session 19: the hand-over when the chip runs empty under the throughput schedule: threshold sweep at the driver's 20 steps and at 60
O=gpurun_out/r05s19; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for b in 512 1024 2048 3072; do for st in 20 60; do
  echo -n "drain below $b, $st steps: "; BIOIK_SOLVE_DRAIN_THROUGHPUT=1 BIOIK_SOLVE_DRAIN_BELOW=$b python bench.py --no-cpu-baseline --timed-only --steps $st --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.3f ms/batch chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done; done; done 2>&1 | tee $O/drain_throughput_sweep.log
