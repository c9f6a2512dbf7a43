#!/bin/bash
# round 3, GPU session 17: the pose-only forms without the one in the pair walk (lib_v_pose3) against with it (lib_u_pose2) and the build before (lib_t_base);
# per-phase cycles of a lone workgroup and of a full chip on the new kernels
O=gpurun_out/s17; mkdir -p $O
export TMPDIR=/tmp
ROUNDS=2 bash tools/step_rate.sh build/lib_u_pose2.so build/lib_v_pose3.so 2>&1 | tee $O/step_rate.log
for rep in 1 2; do for lib in build/lib_t_base.so build/lib_u_pose2.so build/lib_v_pose3.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()}, 'ref-params %.0f tracking %.0f' % (d['reference_parameters']['value'], d['tracking_seeds']['value']))"
done; done 2>&1 | tee $O/bench_ab.log
for c in c2 c3 c4; do BIOIK_SOLVE_REPORT=1 BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py $c $([ $c = c2 ] && echo 1536 || echo 3072); done > $O/phases.log 2>&1
bash tools/lone_probe.sh build/libphase.so >> $O/phases.log 2>&1
