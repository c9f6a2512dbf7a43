#!/bin/bash
for ms in 8 16 32 64; do for b in 4096 16384; do
  v=$(BIOIK_BENCH_MAX_STEPS=$ms BIOIK_BENCH_BATCH=$b python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.2f ms  success %.4f mean steps %.2f' % (d['value'], d['ms_per_step'], d['success_rate'], d['mean_steps_per_solve']))")
  echo "max_steps=$ms batch=$b : $v"
done; done
