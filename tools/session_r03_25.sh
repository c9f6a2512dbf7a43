#!/bin/bash
# round 3, GPU session 25: hand-over of the two-launch solve after K steps on the final kernels (first launch: both species on one wavefront, computed children),
# whose throughput at fixed work is now 30 % above the second launch's mapping (profiles/r03_c2_mappings.log)
O=gpurun_out/s25; mkdir -p $O
export TMPDIR=/tmp
KS="1 2 3 4 6 8 12 16 24" bash tools/two_phase_sweep.sh 2>&1 | tee $O/two_phase.log
