#!/bin/bash
# round 4, GPU session 22: the hand-over machinery with one resident word per XCD, read with device scope: never triggering (cost), and the sweep of the threshold
O=gpurun_out/r04s22; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule throughput --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; }
for rep in 1 2; do for n in 0 1 1024; do
  export BIOIK_SOLVE_DRAIN_BELOW=$n
  echo "drain below $n: isolated $(run 1 24) | 20 steps, 10 in flight $(run 10 20) | 20 steps, 20 in flight $(run 20 20) | 60 steps, 10 in flight $(run 10 60)"
done; done 2>&1 | tee -a $O/drain_sweep.log
