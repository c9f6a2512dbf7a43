#!/bin/bash
# time per step() of one workgroup as a function of how many workgroups share the chip: no query may succeed (dtwist 1e-300),
# so every workgroup runs exactly max_steps steps.  usage: tools/residency_curve.sh [lib]
lib=${1:-bio_ik_amd/libbioik_hip.so}
for thr in 128 256; do for b in 1 256 512 1024 1536 3072; do
  v=$(BIOIK_BENCH_STREAM=0 BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=$b BIOIK_SOLVE_THREADS=$thr python bench.py --timed-only --in-flight 1 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  -> %.1f us per step per workgroup, %.0f steps/ms chip-wide' % (d['ms_per_step'], d['ms_per_step']*1e3/32, $b*32/d['ms_per_step']))")
  echo "threads=$thr batch=$b : $v"
done; done
