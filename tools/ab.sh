#!/bin/bash
# A/B of library variants on the bench workload: tools/ab.sh lib1.so lib2.so ...
for lib in "$@"; do for thr in 128 256; do
  v=$(BIOIK_BENCH_STREAM=0 BIOIK_HIP_LIBRARY=$lib BIOIK_SOLVE_THREADS=$thr python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.2f ms  success %.4f mean steps %.2f' % (d['value'], d['ms_per_step'], d['success_rate'], d['mean_steps_per_solve']))")
  echo "$lib threads=$thr : $v"
done; done
