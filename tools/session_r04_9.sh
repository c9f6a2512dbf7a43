#!/bin/bash
# round 4, GPU session 9: the 128-lane mapping of the latency schedule through k_solve_lean_cl4 (the kernel compiled for that mapping, four wavefronts per SIMD, children
# computed where they are read) against its present kernel k_solve_lean (children kept in columns, 146 registers): fixed work, lone step, and the latency legs of the bench line
O=gpurun_out/r04s9; mkdir -p $O
run() { echo "== $1"; shift; env "$@" SCHEDULE=latency ROUNDS=1 bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so; env "$@" BIOIK_BENCH_BATCH=4096 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_SCHEDULE=latency BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 python bench.py --no-cpu-baseline --timed-only --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  batch 4096: %.3f ms -> %.0f steps/ms' % (d['ms_per_step'], 4096*32/d['ms_per_step']))"
env "$@" BIOIK_BENCH_CONFIGS=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('  bench: value %.0f | lat3 %.0f | one-at-a-time %.0f (%.2f ms) | host entry %.0f | tracking %.0f' % (d['value'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['one_batch_at_a_time']['ms_per_step'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value']))"; }
( run "default (k_solve_lean_cl first step, k_solve_lean the rest)" A=1
  run "k_solve_lean_cl4 for the 128-lane launches" BIOIK_SOLVE_FOUR_WAVES=1 BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2
  run "k_solve_lean_cl (168 registers) for the 128-lane launches" BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 ) 2>&1 | tee $O/latency_kernels.log
