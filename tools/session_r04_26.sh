#!/bin/bash
# round 4, GPU session 26: the resident word prefetched into LDS at the start of a step: the hand-over's bookkeeping on a stream of solves (never triggering: threshold 1),
# the throughput schedule with the hand-over (BIOIK_SOLVE_DRAIN_THROUGHPUT=1), GPU suite
O=gpurun_out/r04s26; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule $3 --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; }
for rep in 1 2 3; do
  echo "throughput, off: 60/10 $(run 10 60 throughput) | 20/10 $(run 10 20 throughput) | 20/20 $(run 20 20 throughput) | isolated $(run 1 24 throughput)"
  echo "throughput, on, never triggers: 60/10 $(BIOIK_SOLVE_DRAIN_THROUGHPUT=1 BIOIK_SOLVE_DRAIN_BELOW=1 run 10 60 throughput)"
  echo "throughput, on, below 1024: 60/10 $(BIOIK_SOLVE_DRAIN_THROUGHPUT=1 run 10 60 throughput) | 20/10 $(BIOIK_SOLVE_DRAIN_THROUGHPUT=1 run 10 20 throughput) | 20/20 $(BIOIK_SOLVE_DRAIN_THROUGHPUT=1 run 20 20 throughput) | isolated $(BIOIK_SOLVE_DRAIN_THROUGHPUT=1 run 1 24 throughput)"
done 2>&1 | tee $O/drain_prefetch.log
echo "latency: isolated $(run 1 24 latency) | three in flight $(run 3 24 latency)" | tee -a $O/drain_prefetch.log
