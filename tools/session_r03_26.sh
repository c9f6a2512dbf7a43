#!/bin/bash
# round 3, GPU session 26: why the half-wavefront mapping's +30 % at fixed work does not show on a stream of real solves: fixed work at 4 / 8 / 16 / 32 steps
# per query, and the whole solve under that mapping with 3 and 6 solves in flight
O=gpurun_out/s26; mkdir -p $O
export TMPDIR=/tmp
H="BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2"
fixed() { for st in 4 8 16 32; do
  v=$(env $1 BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=$st BIOIK_BENCH_BATCH=${2:-3072} python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms -> %.0f steps/ms chip-wide' % (d['ms_per_step'], ${2:-3072}*$st/d['ms_per_step']))")
  echo "$3 batch=${2:-3072} steps=$st : $v"; done; }
{
fixed "X=1" 3072 default_one_launch
fixed "$H" 3072 halves
fixed "X=1" 12288 default_one_launch
fixed "$H" 12288 halves
for nf in 3 6; do
  python bench.py --no-cpu-baseline --timed-only --in-flight $nf --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default, $nf in flight: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
  env $H python bench.py --no-cpu-baseline --timed-only --in-flight $nf --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('halves whole solve, $nf in flight: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
} 2>&1 | tee $O/halves.log
