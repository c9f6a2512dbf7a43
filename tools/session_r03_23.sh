#!/bin/bash
# round 3, GPU session 23: the winner copy skipped when no child made it (lib_y_stay) against the build before (lib_x_base)
# fixed work, bench line with C3 / C4, then the GPU suite on the new build
O=gpurun_out/s23; mkdir -p $O
export TMPDIR=/tmp
ROUNDS=2 bash tools/step_rate.sh build/lib_x_base.so build/lib_y_stay.so 2>&1 | tee $O/step_rate.log
for rep in 1 2; do for lib in build/lib_x_base.so build/lib_y_stay.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log
