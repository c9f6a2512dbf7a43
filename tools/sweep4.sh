#!/bin/bash
# single-query latency probe: small batches, every wave alone on its SIMD
for lib in "$@"; do for b in 64 1024; do for thr in 64 128 256; do
  v=$(BIOIK_HIP_LIBRARY=$lib BIOIK_SOLVE_THREADS=$thr BIOIK_BENCH_BATCH=$b python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.2f ms  mean steps %.2f' % (d['value'], d['ms_per_step'], d['mean_steps_per_solve']))")
  echo "$(basename $lib) batch=$b threads=$thr : $v"
done; done; done
