#!/bin/bash
# round 3, GPU session 19: lane mappings of C3 / C4 on the round's final kernels, with the 128-register build of the computed-children kernel forced
O=gpurun_out/s19; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1)) for k,v in d['configs'].items()})"; }
{
run auto
run auto_again
BIOIK_SOLVE_FOUR_WAVES=1 run auto_w4
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 run t128_cl2
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 run t128_cl2_w4
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=1 run t128_cl1
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=1 BIOIK_SOLVE_FOUR_WAVES=1 run t128_cl1_w4
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_COLUMNLESS=2 run t64_cl2
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 run t64_cl2_w4
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 run t64_sp_cl2
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 run t64_sp_cl2_w4
} 2>&1 | tee $O/mappings.log
