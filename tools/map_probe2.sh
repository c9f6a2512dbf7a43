#!/bin/bash
# the half-wave-species + computed-children mapping on the real bench (launches in flight 3 / 6) and on C3 / C4
real() { python bench.py --no-cpu-baseline --steps 24 --warmup 6 --in-flight $NF 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 in-flight $NF: %.0f solves/s %.2f ms success %.4f | configs' % (d['value'], d['ms_per_step'], d['success_rate']), {k:(round(v['value']),v['success_rate']) for k,v in d.get('configs',{}).items()})"; }
for NF in 3 6; do
real default
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 real t64_sp_columnless_pairs
done
NF=3
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=1 real t64_sp_columnless
BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=0 BIOIK_SOLVE_COLUMNLESS=2 real t64_columnless_pairs
