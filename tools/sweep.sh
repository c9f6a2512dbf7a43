#!/bin/bash
# usage: tools/sweep.sh lib1.so lib2.so ...   (run on the GPU box) — bench value for each library x workgroup shape
for lib in "$@"; do for thr in 128 256; do
  v=$(BIOIK_HIP_LIBRARY=$lib BIOIK_SOLVE_THREADS=$thr python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.2f ms  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
  echo "$(basename $lib) threads=$thr : $v"
done; done
