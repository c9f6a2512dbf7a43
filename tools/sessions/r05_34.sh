#!/bin/bash
# round 5, GPU session 34: the GPU suite and smoke() on the final tree (the branching fixture's device test is new since session 32)
mkdir -p gpurun_out/r05s34; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > gpurun_out/r05s34/gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/r05s34/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
