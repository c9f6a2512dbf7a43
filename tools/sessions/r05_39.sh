#!/bin/bash
# round 5, GPU session 39: the GPU suite on the final tree, last GPU seconds of the round
mkdir -p gpurun_out/r05s39; export TMPDIR=/tmp
( time timeout 60 python -m pytest tests -m gpu -q ) > gpurun_out/r05s39/gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/r05s39/gpu_suite.log
