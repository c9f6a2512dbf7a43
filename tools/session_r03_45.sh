#!/bin/bash
# round 3, GPU session 45: C4's 128-register kernel with the species record read per generation (BIOIK_SOLVE_CL4S=1: 67 -> 31 spilled values)
O=gpurun_out/s45; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1),round(v['success_rate'],3),round(v['roofline']['chip_level_frac'],3)) for k,v in d['configs'].items()})"; }
{
for rep in 1 2 3; do
run cl4
BIOIK_SOLVE_CL4S=1 run cl4s
done
} 2>&1 | tee $O/cl4s.log
