#!/bin/bash
# round 5, GPU session 17: the single-child walk of k_solve_lean_cl4h with the values and the trigonometry of A ops computed ahead of their compositions (fk_walk_ahead): A = 2 / 4 / 8
# against the committed kernel (HEAD~: the plain single-child walk), lone step
O=gpurun_out/r05s17; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for lib in build/ab/lib_r05_ahead2.so build/ab/lib_r05_ahead4.so build/ab/lib_r05_ahead8.so; do
  echo -n "$lib: "; BIOIK_HIP_LIBRARY=$lib timeout 120 python tools/lone_call_overhead.py 1 2>&1 | grep "islands 1:"
done; done 2>&1 | tee $O/walk_ahead.log
SMALL_SIZES=1,16,64 timeout 300 python tools/small_batches.py "ahead4_auto:;islands=0" 2>&1 | grep -v amdgpu | tee -a $O/walk_ahead.log
( time timeout 600 python -m pytest tests -m gpu -q -x -k "helper or trajectory or islands" ) 2>&1 | grep -E "passed|failed" | tee -a $O/walk_ahead.log
