#!/bin/bash
# round 5, GPU session 10: timeout counted from the call's submission; device properties instead of literals; mimic chains resolved; GPU suite + the isolated-call / small-batch figures
O=gpurun_out/r05s10; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log; grep -n "^E " $O/gpu_suite.log | head
BIOIK_SOLVE_REPORT=1 python tools/lone_call_overhead.py 1 2>&1 | grep "islands [18]:"
