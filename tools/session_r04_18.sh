#!/bin/bash
# round 4, GPU session 18: the throughput schedule's dense kernel with its stragglers handed to the latency mapping after K steps (BIOIK_SOLVE_DENSE_HANDOVER=K), on an
# isolated call (one solve after the other), three and six in flight at the driver's step count; against the latency schedule's own isolated call
O=gpurun_out/r04s18; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule $1 --in-flight $2 --steps $3 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms success %.4f' % (d['value'], d['ms_per_step'], d.get('success_rate', -1)))"; }
echo "latency schedule, in flight 1: $(run latency 1 24)" | tee -a $O/handover_sweep.log
for inf in 1 3 10; do for k in 0 4 6 8 10 12 16 24; do
  echo "throughput schedule, in flight $inf, hand-over after $k: $(BIOIK_SOLVE_DENSE_HANDOVER=$k run throughput $inf $([ $inf = 1 ] && echo 24 || echo 20))"
done; done 2>&1 | tee -a $O/handover_sweep.log
