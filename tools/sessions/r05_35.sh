#!/bin/bash
# round 5, GPU session 35: the GPU suite and smoke() on the final tree (new since session 34: BIOIK_COMPILE_EXACT, the unfolded joint program, and its device test)
mkdir -p gpurun_out/r05s35; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > gpurun_out/r05s35/gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/r05s35/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
