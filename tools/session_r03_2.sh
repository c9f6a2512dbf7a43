#!/bin/bash
# round 3, GPU session 2: GPU parity suite on the new library (all of K1-K3), A/B of library builds (fixed-work step rate, bench line)
O=gpurun_out/s2; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
tail -3 $O/gputests.log
bash tools/step_rate.sh build/lib_b.so build/lib_f_all.so build/lib_g_k13.so build/lib_h_k2.so build/lib_i_nosink.so build/lib_j_licm.so build/lib_b.so build/lib_f_all.so > $O/step_rate.log 2>&1
cat $O/step_rate.log
for lib in build/lib_b.so build/lib_f_all.so build/lib_i_nosink.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1)), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()})"
done 2>&1 | tee $O/bench_ab.log
