#!/bin/bash
# round 4, GPU session 25: small batches under the latency schedule, one call at a time: the launcher's choice (<= 768 units: 256 lanes, children in columns)
# against k_solve_lean_cl4 forced (128 lanes, computed children in pairs) and the dense kernel (throughput schedule)
O=gpurun_out/r04s25; mkdir -p $O
export TMPDIR=/tmp
run() { BIOIK_BENCH_BATCH=$1 timeout 120 python bench.py --timed-only --in-flight 1 --no-cpu-baseline --schedule $2 --steps 12 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
for b in 1 16 64 256 512 768 1024 1536 2048; do
  echo "batch $b: launcher $(run $b latency) | cl4 forced $(BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 run $b latency) | dense $(run $b throughput)"
done 2>&1 | tee $O/small_batches.log
