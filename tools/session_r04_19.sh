#!/bin/bash
# round 4, GPU session 19: the driver's command (20 timed steps) against the number of solves in flight
O=gpurun_out/r04s19; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule throughput --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; }
for rep in 1 2; do for inf in 4 5 6 8 10 12 16 20; do
  echo "20 steps, in flight $inf: $(run $inf 20)"
done; done 2>&1 | tee -a $O/inflight_sweep.log
for inf in 6 10 20; do echo "60 steps, in flight $inf: $(run $inf 60)"; done 2>&1 | tee -a $O/inflight_sweep.log
