#!/bin/bash
# round 5, GPU session 33: a 20 000-case parity soak on the final tree (the draw includes the two fixtures with several secondary goals and a third of the cases with the sort of all children)
mkdir -p gpurun_out/r05s33; export TMPDIR=/tmp
( time timeout 420 python tools/fuzz_parity.py 20000 97531 ) > gpurun_out/r05s33/fuzz_20000.log 2>&1; grep -v " ok$" gpurun_out/r05s33/fuzz_20000.log | tail -6
