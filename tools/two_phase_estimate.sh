#!/bin/bash
# what a two-phase launch could give (not built): phase 1 = the half-wave-species mapping with computed children for K steps, measured as the
# bench line with max_steps = K; phase 2 = the stragglers on the default mapping (estimated from the step histogram, not measured)
for K in 8 12 16 24 32; do
for m in default sp; do
  if [ $m = sp ]; then export BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2; else unset BIOIK_SOLVE_THREADS BIOIK_SOLVE_SPECIES_PARALLEL BIOIK_SOLVE_COLUMNLESS; fi
  BIOIK_BENCH_MAX_STEPS=$K python bench.py --no-cpu-baseline --steps 24 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$K $m: %.2f ms per batch (three in flight), one at a time %.2f ms, success %.4f mean steps %.2f' % (d['ms_per_step'], d['one_batch_at_a_time']['ms_per_step'], d['success_rate'], d['mean_steps_per_solve']))"
done; done
