#!/bin/bash
# do the solves in flight have to be the SAME batch for the two-launch form to win?  (BIOIK_BENCH_DISTINCT=1: one batch of queries per stream)
for DIS in "" 1; do for TP in 0 1 init; do
BIOIK_BENCH_DISTINCT=$DIS BIOIK_SOLVE_TWO_PHASE=$TP python bench.py --no-cpu-baseline --timed-only --steps 48 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('distinct=${DIS:-0} two_phase=$TP: %.2f ms per batch (%.0f solves/s counted on stream 0s batch)' % (d['ms_per_step'], d['value']))"
done; done
