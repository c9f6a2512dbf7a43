#!/bin/bash
# lane mappings of the C2 workload at fixed work (no query may succeed, 32 steps each): chip-wide step rate per forced mapping,
# then the real bench line (three launches in flight) for the interesting ones.  usage: tools/map_probe.sh
run() { v=$(BIOIK_SOLVE_REPORT=1 BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=$B python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/tmp/err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms, %.0f steps/ms chip-wide' % (d['ms_per_step'], $B*32/d['ms_per_step']))"); echo "$1 batch=$B: $v | $(grep bioik /tmp/err | sort | uniq -c | sort -rn | head -1 | sed 's/.*lanes/lanes/')"; }
B=3072
run default
BIOIK_SOLVE_THREADS=64 run t64
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_COLUMNLESS=2 run t64_columnless_pairs
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_COLUMNLESS=1 run t64_columnless
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 run t64_sp_columnless_pairs
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=1 run t64_sp_columnless
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 run t128_columnless_pairs
real() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value']))"; }
real default
BIOIK_SOLVE_THREADS=64 real t64
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 real t64_sp_columnless_pairs
