#!/bin/bash
# species on the two halves of one wavefront (64 lanes, BIOIK_SOLVE_SPECIES_PARALLEL=1) against the default mapping
for cfg in "16 linear 512" "16 exact 512" "32 exact 256" "64 exact 128" "128 exact 64"; do set -- $cfg
  for map in default half; do
    if [ $map = half ]; then export BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1; else unset BIOIK_SOLVE_THREADS BIOIK_SOLVE_SPECIES_PARALLEL; fi
    v=$(BIOIK_BENCH_STREAM=0 BIOIK_BENCH_POP=$1 BIOIK_BENCH_FK=$2 BIOIK_BENCH_MAX_STEPS=$3 python bench.py --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s (three in flight) %.2f ms/batch; one at a time %.2f ms; success %.4f mean steps %.2f' % (d['value'], d['ms_per_step'], d['one_batch_at_a_time']['ms_per_step'], d['success_rate'], d['mean_steps_per_solve']))")
    echo "pop=$1 fk=$2 max_steps=$3 mapping=$map : $v"
  done
done
