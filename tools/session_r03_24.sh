#!/bin/bash
# round 3, GPU session 24: C2 under the computed-children mappings on the final kernels (fixed work and bench line)
O=gpurun_out/s24; mkdir -p $O
export TMPDIR=/tmp
lib=bio_ik_amd/libbioik_hip.so
{
for r in 1 2; do
echo "== default"; bash tools/step_rate.sh $lib | grep 3072
echo "== columnless 2 (128 lanes)"; BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 ROUNDS=1 bash tools/step_rate.sh $lib | grep 3072
echo "== columnless 2, four wavefronts"; BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 ROUNDS=1 bash tools/step_rate.sh $lib | grep 3072
echo "== halves, columnless 2"; BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 ROUNDS=1 bash tools/step_rate.sh $lib | grep 3072
echo "== halves, columnless 2, four wavefronts"; BIOIK_SOLVE_THREADS=64 BIOIK_SOLVE_SPECIES_PARALLEL=1 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_FOUR_WAVES=1 ROUNDS=1 bash tools/step_rate.sh $lib | grep 3072
done
} 2>&1 | tee $O/c2_mappings.log
