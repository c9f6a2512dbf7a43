#!/bin/bash
# round 5, GPU session 6: the wide kernel (k_solve_lean_wide) on launches that cannot fill the chip, against k_solve_lean_cl4; with islands
O=gpurun_out/r05s06; mkdir -p $O
export TMPDIR=/tmp
{
python tools/lone_call_overhead.py 1 2>&1 | grep -v amdgpu.ids | grep "islands [18]:"
BIOIK_SOLVE_WIDE=0 python tools/lone_call_overhead.py 1 2>&1 | grep -v amdgpu.ids | grep "islands [18]:"
SMALL_SIZES=1,16,64,128,256,512,1024 python tools/small_batches.py "cl4:BIOIK_SOLVE_WIDE=0" "wide256:BIOIK_SOLVE_WIDE=256" "wide1024:BIOIK_SOLVE_WIDE=1024" \
   "cl4_islands_auto8:BIOIK_SOLVE_WIDE=0;islands=-8" "wide256_islands_auto8:BIOIK_SOLVE_WIDE=256;islands=-8" "wide1024_islands_auto8:BIOIK_SOLVE_WIDE=1024;islands=-8" "wide4096_islands_auto16:BIOIK_SOLVE_WIDE=4096;islands=-16" 2>&1 | grep -v amdgpu.ids
} | tee $O/wide_kernel.log
