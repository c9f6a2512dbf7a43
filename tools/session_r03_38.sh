#!/bin/bash
# round 3, GPU session 38: the dense kernel compiled for single-wavefront workgroups (__launch_bounds__(64, 3): BIOIK_SOLVE_CL64; (64, 4): BIOIK_SOLVE_CL64W4)
O=gpurun_out/s38; mkdir -p $O
export TMPDIR=/tmp
{
for rep in 1 2 3; do for v in "X=1" "BIOIK_SOLVE_CL64=1" "BIOIK_SOLVE_CL64W4=1"; do
  env $v python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v bench: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done; done
for v in "X=1" "BIOIK_SOLVE_CL64=1" "BIOIK_SOLVE_CL64W4=1"; do
  r=$(env $v BIOIK_BENCH_SCHEDULE=throughput BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=3072 python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms -> %.0f steps/ms chip-wide' % (d['ms_per_step'], 3072*32/d['ms_per_step']))")
  echo "$v fixed work, dense mapping: $r"
done
} 2>&1 | tee $O/cl64.log
