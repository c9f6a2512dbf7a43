#!/bin/bash
# round 5, GPU session 23: k_solve_lean_cl4h for small launches of serial chains WITH secondary goals (a 7-joint arm + MinimalDisplacementGoal, the 31-joint chain + AvoidJointLimitsGoal)
mkdir -p gpurun_out; export TMPDIR=/tmp
{
SMALL_PROBLEM=c2sec SMALL_SIZES=1,16,64,256 python tools/small_batches.py "arm_mindisp_rules_of_round4:BIOIK_SOLVE_HELPED=0;islands=0" "arm_mindisp_helped:;islands=0" 2>&1 | grep -v amdgpu
SMALL_PROBLEM=c4 SMALL_POP=512 SMALL_STEPS=32 SMALL_REPS=12 SMALL_SIZES=1,16,64,256 python tools/small_batches.py "snake31_rules_of_round4:BIOIK_SOLVE_HELPED=0;islands=0" "snake31_helped:;islands=0" 2>&1 | grep -v amdgpu
} | tee gpurun_out/r05s23_helped_secondary.log
( time python -m pytest tests -m gpu -q -x ) > gpurun_out/r05s23_gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/r05s23_gpu_suite.log; grep -n "^E " gpurun_out/r05s23_gpu_suite.log | head -5
python tools/fuzz_parity.py 3000 4321 2>&1 | tail -1
