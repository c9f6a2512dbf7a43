#!/bin/bash
# round 5, GPU session 36: the device test of the unfolded joint program once more (whole solves with LookAt / Cone goals left to the host simulator: another acos), and the whole suite
mkdir -p gpurun_out/r05s36; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > gpurun_out/r05s36/gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/r05s36/gpu_suite.log
