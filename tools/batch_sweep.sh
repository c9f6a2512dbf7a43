#!/bin/bash
# throughput vs batch size and lanes per query (is the launch tail-bound?)
for cfg in "1024 128" "1024 256" "4096 128" "4096 256" "16384 128" "65536 128"; do set -- $cfg
  v=$(BIOIK_BENCH_BATCH=$1 BIOIK_SOLVE_THREADS=$2 python bench.py --timed-only --in-flight 1 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.2f ms  success %.4f mean steps %.2f' % (d['value'], d['ms_per_step'], d['success_rate'], d['mean_steps_per_solve']))")
  echo "batch=$1 threads=$2 : $v"
done
