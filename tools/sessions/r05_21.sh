#!/bin/bash
# round 5, GPU session 21: from how many units on the latency schedule should start under the dense kernel and hand its stragglers to k_solve_lean_cl4h (round 4, with k_solve_lean_cl4 as the consumer: 3072)
O=gpurun_out/r05s21; mkdir -p $O
export TMPDIR=/tmp
for n in 1280 1536 2048 3072; do for m in 1025 100000; do
  echo -n "n $n, dense first from $m units: "; BIOIK_SOLVE_DRAIN_MIN_UNITS=$m BIOIK_BENCH_BATCH=$n python bench.py --timed-only --in-flight 1 --schedule latency --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.3f ms per call' % (d['value'], d['ms_per_step']))"
done; done 2>&1 | tee $O/drain_min_units.log
