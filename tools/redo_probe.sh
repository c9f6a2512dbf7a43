#!/bin/bash
for lib in ab/a_head.so ab/flag_order.so; do for TP in 0 1 init; do
BIOIK_HIP_LIBRARY=$lib BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_TWO_PHASE=$TP python bench.py --no-cpu-baseline --timed-only --steps 48 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib two_phase=$TP: %.0f solves/s %.2f ms per batch' % (d['value'], d['ms_per_step']))"
done; done
