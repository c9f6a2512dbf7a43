#!/bin/bash
# does the number of hardware queues HIP multiplexes streams onto (GPU_MAX_HW_QUEUES, default 4) decide how well solves in flight overlap?
for Q in 4 8; do for TP in 0 1; do for NF in 3 4 6; do
GPU_MAX_HW_QUEUES=$Q BIOIK_SOLVE_TWO_PHASE=$TP python bench.py --no-cpu-baseline --timed-only --steps 48 --warmup 6 --in-flight $NF 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hw queues $Q two_phase=$TP in flight $NF: %.0f solves/s %.2f ms per batch' % (d['value'], d['ms_per_step']))"
done; done; done
