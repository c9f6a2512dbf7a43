#!/bin/bash
lib=$1
for cfg in "16 linear 512 64" "16 linear 512 128" "32 linear 256 64" "64 linear 128 128" "128 linear 64 256" "16 exact 512 64" "64 exact 128 128"; do set -- $cfg
  v=$(BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_POP=$1 BIOIK_BENCH_FK=$2 BIOIK_BENCH_MAX_STEPS=$3 BIOIK_SOLVE_THREADS=$4 python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.2f ms  success %.4f mean steps %.2f' % (d['value'], d['ms_per_step'], d['success_rate'], d['mean_steps_per_solve']))")
  echo "pop=$1 fk=$2 max_steps=$3 threads=$4 : $v"
done
