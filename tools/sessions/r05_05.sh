#!/bin/bash
# round 5, GPU session 5: calls that cannot fill the chip -- lane mappings and islands (what exists in the tree, before anything is built)
O=gpurun_out/r05s05; mkdir -p $O
export TMPDIR=/tmp
python tools/small_batches.py "default:" "lanes256_single:BIOIK_SOLVE_THREADS=256,BIOIK_SOLVE_COLUMNLESS=1" "lanes256_columns:BIOIK_SOLVE_THREADS=256" \
   "islands2:;islands=2" "islands4:;islands=4" "islands8:;islands=8" "islands_auto16:;islands=-16" "lanes256_single_islands4:BIOIK_SOLVE_THREADS=256,BIOIK_SOLVE_COLUMNLESS=1;islands=4" 2>&1 | grep -v amdgpu.ids | tee $O/small_batches.log
