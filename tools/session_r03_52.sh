#!/bin/bash
# round 3, GPU session 52: the candidates of session 51 on the tracking seeds (seed = target + N(0, 0.1 rad)) as well: BIOIK_SCHEDULE_LATENCY, three in flight
O=gpurun_out/s52; mkdir -p $O
export TMPDIR=/tmp BIOIK_BENCH_CONFIGS=0
for v in "1 0" "4 1" "8 1" "12 1" "16 1"; do set -- $v
  if [ $2 = 1 ]; then export BIOIK_SOLVE_CL64W4=1; else unset BIOIK_SOLVE_CL64W4; fi
  r=$(BIOIK_SOLVE_TWO_PHASE=$1 timeout 300 python bench.py --no-cpu-baseline --schedule latency --in-flight 3 --steps 24 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('three in flight %.0f  one at a time %.0f  tracking seeds %.0f (mean steps %.2f)' % (d['value'], d['one_batch_at_a_time']['value'], d['tracking_seeds']['value'], d['tracking_seeds']['mean_steps_per_solve']))")
  echo "hand-over after $1, first launch $( [ $2 = 1 ] && echo '128-register' || echo '168-register' ): $r"
done 2>&1 | tee $O/handover_tracking.log
