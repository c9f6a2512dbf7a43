#!/bin/bash
# round 3, GPU session 21: the joint walk with a wavefront per species (128 lanes; C4, and C3 under that mapping) against the launcher's choices
O=gpurun_out/s21; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: configs', {k:(round(v['value']),round(v['ms_per_step'],1),round(v['success_rate'],3),round(v['roofline']['chip_level_frac'],3)) for k,v in d['configs'].items()})"; }
{
for rep in 1 2 3; do
run auto
BIOIK_SOLVE_JOINT_128=1 run auto_joint128
BIOIK_SOLVE_JOINT_128=1 BIOIK_SOLVE_THREE_WAVES=1 run auto_joint128_w3
done
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_JOINT_128=1 run t128_cl2_joint
BIOIK_SOLVE_THREADS=128 BIOIK_SOLVE_COLUMNLESS=2 BIOIK_SOLVE_JOINT_128=1 BIOIK_SOLVE_FOUR_WAVES=1 run t128_cl2_joint_w4
} 2>&1 | tee $O/joint128.log
