#!/bin/bash
# round 5, GPU session 12: the launcher's measured mapping choice (solve_dispatch): the table per configuration, rules against choice; GPU suite
O=gpurun_out/r05s12; mkdir -p $O
export TMPDIR=/tmp
python tools/autotune_probe.py 2>&1 | grep -v "amdgpu.ids\|joint program\|\[bioik\] solve:\|\[bioik\] launch:" | tee $O/autotune.log
( time python -m pytest tests -m gpu -q -x ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log; grep -n "^E " $O/gpu_suite.log | head -5
