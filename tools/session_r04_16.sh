#!/bin/bash
# round 4, GPU session 16: the critical path of ONE query (nobody else on the chip) per phase, under the latency schedule's kernel and under 256 lanes
O=gpurun_out/r04s16; mkdir -p $O
export BIOIK_HIP_LIBRARY=build/ab/libphase.so BIOIK_SOLVE_REPORT=1
( python tools/phase_probe_config.py c2 1 latency; BIOIK_SOLVE_THREADS=256 python tools/phase_probe_config.py c2 1 latency; python tools/phase_probe_config.py c2 1 throughput; python tools/phase_probe_config.py c2 64 latency ) 2>&1 | grep -v "amdgpu.ids" | tee $O/phases.log
