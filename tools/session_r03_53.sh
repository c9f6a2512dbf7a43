#!/bin/bash
# round 3, GPU session 53: the GPU suite and smoke() on the final tree (what the driver runs at round end)
O=gpurun_out/s53; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -4 $O/gpu_suite.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"
tail -4 $O/smoke.log
