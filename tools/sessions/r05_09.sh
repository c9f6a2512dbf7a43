#!/bin/bash
# round 5, GPU session 9: the dense kernel under the register budget of FIVE wavefronts per SIMD (98 VGPRs, 84 spilled values, 128 B of scratch) against the product's four
O=gpurun_out/r05s09; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for lib in bio_ik_amd/libbioik_hip.so build/ab/lib_r05_dense5.so; do
  v=$(BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.3f ms/batch chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))")
  echo "$lib : $v"
done; done 2>&1 | tee $O/ab_dense5.log
SCHEDULE=throughput ROUNDS=1 bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so build/ab/lib_r05_dense5.so 2>&1 | grep "batch=3072" | tee -a $O/ab_dense5.log
