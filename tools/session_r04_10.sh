#!/bin/bash
# round 4, GPU session 10: lib_a7 = the latency schedule's 128-lane launches through k_solve_lean_cl4; full bench legs, the dense hand-over to it
# (BIOIK_SOLVE_DENSE_HANDOVER), the two-launch hand-over step (BIOIK_SOLVE_TWO_PHASE), GPU suite
O=gpurun_out/r04s10; mkdir -p $O
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1: value %.0f (%.2f ms, chip %.3f) | lat3 %.0f | one-at-a-time %.0f (%.2f ms) | host entry %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['one_batch_at_a_time']['ms_per_step'], d['host_pointer_entry']['solves_per_s'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value']))"; }
( for lib in build/ab/lib_a6.so build/ab/lib_a7.so; do BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "$lib driver-cmd"; done
for K in 16 24 32; do BIOIK_SOLVE_DENSE_HANDOVER=$K BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=build/ab/lib_a7.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "lib_a7 dense handover=$K"; done
for TP in 0 2 4 8; do BIOIK_SOLVE_TWO_PHASE=$TP BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=build/ab/lib_a7.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "lib_a7 two-phase=$TP"; done ) 2>&1 | tee $O/bench_ab.log
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -5 | tee $O/gpu_suite.log
