#!/bin/bash
# A/B of library variants with the bench's default issue pattern (two batches in flight): tools/ab_pipe.sh lib1.so lib2.so ...
for rep in 1 2; do for lib in "$@"; do
  v=$(BIOIK_BENCH_STREAM=0 BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.2f ms/batch (one at a time %.2f ms)' % (d['value'], d['ms_per_step'], d['one_batch_at_a_time']['ms_per_step']))")
  echo "$lib : $v"
done; done
