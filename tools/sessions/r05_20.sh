#!/bin/bash
# round 5, GPU session 20: the throughput schedule hands its stragglers over by default (threshold 512 wavefronts, stragglers under k_solve_lean_cl4h): GPU suite, the driver's command, the default command
O=gpurun_out/r05s20; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log; grep -n "^E " $O/gpu_suite.log | head -5
( time python tools/fuzz_parity.py 1500 1212 ) 2>&1 | grep -v " ok$" | tail -3
for st in "20 5" "60 3"; do set -- $st; python bench.py --no-cpu-baseline --timed-only --steps $1 --warmup $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $1: %.0f solves/s  %.3f ms/batch chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"; done
