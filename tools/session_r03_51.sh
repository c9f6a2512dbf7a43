#!/bin/bash
# round 3, GPU session 51: an isolated call (one solve after the other) and three in flight under BIOIK_SCHEDULE_LATENCY:
# hand-over after K steps x the first launch's kernel (168-register computed-children kernel / its 128-register build: 4096 wavefronts = the whole
# batch resident at once)
O=gpurun_out/s51; mkdir -p $O
export TMPDIR=/tmp
for inf in 1 3; do for cl in 0 1; do for k in 1 4 8 12 16 24; do
  if [ $cl = 1 ]; then export BIOIK_SOLVE_CL64W4=1; else unset BIOIK_SOLVE_CL64W4; fi
  v=$(BIOIK_SOLVE_TWO_PHASE=$k timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule latency --in-flight $inf --steps 24 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "in flight $inf, first launch $( [ $cl = 1 ] && echo '128-register' || echo '168-register' ), hand-over after $k: $v"
done; done; done 2>&1 | tee $O/handover_sweep.log
