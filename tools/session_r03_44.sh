#!/bin/bash
# round 3, GPU session 44: C3's joint-walk kernel under the four-wavefront budget with the species record per generation (BIOIK_SOLVE_CLJ4=1)
O=gpurun_out/s44; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1),round(v['roofline']['chip_level_frac'],3)) for k,v in d['configs'].items()})"; }
{
for rep in 1 2 3; do
run clj
BIOIK_SOLVE_CLJ4=1 run clj4
done
} 2>&1 | tee $O/clj4.log
