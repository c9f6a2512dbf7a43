#!/bin/bash
# round 3, GPU session 27: solves in flight x hardware queues for the default two-launch solve, a late hand-over (K=12) and the whole solve under the
# half-wavefront mapping -- does the +27 % of that mapping at fixed work need more work in flight to show on real solves?
O=gpurun_out/s27; mkdir -p $O
export TMPDIR=/tmp
H="BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2"
one() { env $2 python bench.py --no-cpu-baseline --timed-only --in-flight $3 --steps 48 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1, $3 in flight: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"; }
{
for q in 4 8; do
export GPU_MAX_HW_QUEUES=$q
echo "== GPU_MAX_HW_QUEUES=$q"
for nf in 3 6 8 12; do
  one "default (hand-over after 1)" "X=1" $nf
  one "hand-over after 12" "BIOIK_SOLVE_TWO_PHASE=12" $nf
  one "halves whole solve" "$H" $nf
done
done
} 2>&1 | tee $O/inflight.log
