#!/bin/bash
# round 4, GPU session 24: the latency schedule with the dense kernel first and its stragglers handed to k_solve_lean_cl4 when the chip runs empty (the default now)
# against k_solve_lean_cl4 alone (BIOIK_SOLVE_DRAIN_BELOW=0): an isolated call, three in flight; GPU suite; the bench line
O=gpurun_out/r04s24; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
run() { timeout 120 python bench.py --timed-only --no-cpu-baseline --schedule latency --in-flight $1 --steps $2 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do for n in 0 512 1024 1536; do
  export BIOIK_SOLVE_DRAIN_BELOW=$n
  echo "latency schedule, drain below $n: isolated $(run 1 24) | three in flight $(run 3 24)"
done; done 2>&1 | tee -a $O/latency_drain.log
unset BIOIK_SOLVE_DRAIN_BELOW
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('driver cmd: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/latency_drain.log
