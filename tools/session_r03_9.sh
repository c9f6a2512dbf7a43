#!/bin/bash
# round 3, GPU session 9: only k_solve_lean (the second launch of a C2 solve) under the four-wavefront budget
O=gpurun_out/s9; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for lib in build/lib_k_coop.so build/lib_o_w3.so build/lib_o_lean4.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log
for lib in build/lib_o_w3.so build/lib_o_lean4.so; do for n in 64 512 1024 2048 4096 16384; do
  BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_BATCH=$n BIOIK_BENCH_CONFIGS=0 BIOIK_BENCH_STREAM=0 python bench.py --no-cpu-baseline --timed-only --in-flight 1 --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib one call of $n queries: %.2f ms, %.0f solves/s' % (d['ms_per_step'], d['value']))"
done; done 2>&1 | tee $O/batch_sizes.log
