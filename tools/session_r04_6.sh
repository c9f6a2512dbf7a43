#!/bin/bash
# round 4, GPU session 6: per-phase shader cycles (-DBIOIK_PHASE_TIMING build of the present sources) of C2 under the throughput schedule (dense kernel), full chip and lone,
# of the reference's own parameters (pop 16, linear), of C3 and C4
O=gpurun_out/r04s6; mkdir -p $O
export BIOIK_HIP_LIBRARY=build/ab/libphase.so BIOIK_SOLVE_REPORT=1
( python tools/phase_probe_config.py c2 4096 throughput; python tools/phase_probe_config.py c2 1 throughput; python tools/phase_probe_config.py ref 4096; python tools/phase_probe_config.py ref 1; python tools/phase_probe_config.py c3 3072; python tools/phase_probe_config.py c4 2048 ) 2>&1 | grep -v "amdgpu.ids" | tee $O/phases.log
