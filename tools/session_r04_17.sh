#!/bin/bash
# round 4, GPU session 17: the critical path of ONE query under k_solve_lean_cl4 (forced: 128 lanes, computed children in pairs), per phase
O=gpurun_out/r04s17; mkdir -p $O
export BIOIK_HIP_LIBRARY=build/ab/libphase.so BIOIK_SOLVE_REPORT=1
( BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 python tools/phase_probe_config.py c2 1 latency; BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 python tools/phase_probe_config.py c2 256 latency; python tools/phase_probe_config.py c2 4096 latency ) 2>&1 | grep -v "amdgpu.ids" | tee $O/phases.log
