#!/bin/bash
for TP in 0 1; do for NF in 1 2 3 4 6; do
BIOIK_SOLVE_TWO_PHASE=$TP python bench.py --no-cpu-baseline --timed-only --steps 48 --warmup 6 --in-flight $NF 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two_phase=$TP in flight $NF: %.0f solves/s %.2f ms per batch, per-solve kernels %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
