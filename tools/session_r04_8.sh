#!/bin/bash
# round 4, GPU session 8: k_solve_lean_lin (small populations, linearised phenotypes: the reference's own parameters) -- lib_a5 (k_solve_lean, 146 registers, twelve
# queries per CU) against lib_a6 (the kernel compiled for that mapping, 121 registers, sixteen per CU): fixed work and the bench line's reference_parameters leg; GPU suite
O=gpurun_out/r04s8; mkdir -p $O
sed -n '/^probe()/,/^}/p' tools/session_r04_7.sh > /tmp/probe.sh; source /tmp/probe.sh
for lib in build/ab/lib_a5.so build/ab/lib_a6.so; do for n in 4096 8192; do BIOIK_HIP_LIBRARY=$lib probe $n 16 2>&1 | grep -v amdgpu.ids | sed "s|^|$lib |"; done; BIOIK_HIP_LIBRARY=$lib probe 4096 32 2>&1 | grep -v amdgpu.ids | sed "s|^|$lib |"; done | tee $O/fixed_work.log
for lib in build/ab/lib_a5.so build/ab/lib_a6.so; do
BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['reference_parameters']
print('$lib: value %.0f | reference_parameters %.0f solves/s (%.2f ms, success %.4f, %.1f steps) | tracking %.0f' % (d['value'], r['value'], r['ms_per_step'], r['success_rate'], r['mean_steps_per_solve'], d['tracking_seeds']['value']))" | tee -a $O/bench_ab.log
done
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -5 | tee $O/gpu_suite.log
