#!/bin/bash
# round 3, GPU session 6: four wavefronts per SIMD (128 VGPRs) now that the hot loops fit; one- vs two-launch solve on the new kernels
O=gpurun_out/s6; mkdir -p $O
export TMPDIR=/tmp
ROUNDS=2 bash tools/step_rate.sh build/lib_k_coop.so build/lib_l_w4.so build/lib_l_w4_nocoop.so > $O/step_rate.log 2>&1
cat $O/step_rate.log
for lib in build/lib_k_coop.so build/lib_l_w4.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()})"
done 2>&1 | tee $O/bench_ab.log
for k in 0 1 2; do
  BIOIK_SOLVE_TWO_PHASE=$k BIOIK_BENCH_CONFIGS=0 BIOIK_BENCH_STREAM=0 python bench.py --no-cpu-baseline --steps 36 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two_phase=$k: %.0f solves/s %.2f ms | one at a time %.0f' % (d['value'], d['ms_per_step'], d['one_batch_at_a_time']['value']))"
done 2>&1 | tee $O/two_phase.log
BIOIK_BENCH_CONFIGS=0 BIOIK_BENCH_STREAM=0 python bench.py --no-cpu-baseline --steps 36 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two_phase=auto: %.0f solves/s %.2f ms | one at a time %.0f' % (d['value'], d['ms_per_step'], d['one_batch_at_a_time']['value']))" | tee -a $O/two_phase.log
