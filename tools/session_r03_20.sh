#!/bin/bash
# round 3, GPU session 20: the joint walk of both species' children (k_solve_lean_clj; C3) against one half of the wavefront per species (BIOIK_SOLVE_NO_JOINT=1)
O=gpurun_out/s20; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1),round(v['success_rate'],3),round(v['roofline']['chip_level_frac'],3)) for k,v in d['configs'].items()})"; }
{
for rep in 1 2 3; do
BIOIK_SOLVE_NO_JOINT=1 run halves
run joint
done
} 2>&1 | tee $O/joint.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/gpu_suite.log
( time timeout 900 python tools/fuzz_parity.py 600 9 ) > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
