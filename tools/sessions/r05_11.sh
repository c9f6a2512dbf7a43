#!/bin/bash
# round 5, GPU session 11: floating / planar joints anywhere on the chains, mimic chains; GPU suite; C3 success rates at the tests' budgets
O=gpurun_out/r05s11; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log; grep -n "^E " $O/gpu_suite.log | head
python tools/c3_rates.py 2>&1 | grep -v amdgpu | tee $O/c3_rates.log
