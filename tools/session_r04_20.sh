#!/bin/bash
# round 4, GPU session 20: hand-over of the throughput schedule's stragglers when the chip runs empty (BIOIK_SOLVE_DRAIN_BELOW=N resident workgroups):
# an isolated call, the driver's command (20 steps, 10 and 20 in flight), the steady state (60 steps)
O=gpurun_out/r04s20; mkdir -p $O
export TMPDIR=/tmp

run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule throughput --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f success %.4f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d.get('success_rate', -1)))"; }
for n in 0 512 1024 1536 2048; do
  export BIOIK_SOLVE_DRAIN_BELOW=$n
  echo "drain below $n: isolated $(run 1 24) | 20 steps, 10 in flight $(run 10 20) | 20 steps, 20 in flight $(run 20 20) | 60 steps, 10 in flight $(run 10 60)"
done 2>&1 | tee -a $O/drain_sweep.log
for m in 8; do
  export BIOIK_SOLVE_DRAIN_BELOW=1536 BIOIK_SOLVE_DRAIN_MIN_STEPS=$m
  echo "drain below 1536, min steps $m: isolated $(run 1 24) | 20 steps, 10 in flight $(run 10 20) | 20 steps, 20 in flight $(run 20 20)"
done 2>&1 | tee -a $O/drain_sweep.log
