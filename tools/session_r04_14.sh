#!/bin/bash
# round 4, GPU session 14: per-phase shader cycles of C3 under k_solve_lean_clj4 with the pre-selection split into scoring | sort | draw
O=gpurun_out/r04s14; mkdir -p $O
export BIOIK_HIP_LIBRARY=build/ab/libphase.so BIOIK_SOLVE_REPORT=1
( python tools/phase_probe_config.py c3 3072; python tools/phase_probe_config.py c4 2048 ) 2>&1 | grep -v "amdgpu.ids" | tee $O/phases.log
