#!/bin/bash
# round 3, GPU session 32: per-phase cycles of a C2 step under the dense mapping of the throughput schedule (both species on one wavefront), full chip and lone
O=gpurun_out/s32; mkdir -p $O
export TMPDIR=/tmp
H="BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2"
{
env $H BIOIK_SOLVE_REPORT=1 BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py c2 3072
env $H BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py c2 1
BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py c2 1536
} 2>&1 | grep -v "amdgpu.ids\|Warning\|getlimits\|machar" | tee $O/phases_dense.log
