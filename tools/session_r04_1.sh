#!/bin/bash
# round 4, GPU session 1: the GPU suite on the kernels compiled for one mapping (solve_body<.., FIXED>), then A/B of library builds on fixed work (throughput
# schedule) and on the driver's bench command: lib_r03 = round 3's final sources, lib_a1 = FIXED kernels without register spills, lib_a2 = + the clamp as
# two instructions; BIOIK_SOLVE_DENSE_HANDOVER = a throughput solve hands its stragglers to the latency mapping after K steps
O=gpurun_out/r04s1; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1
tail -4 $O/gpu_suite.log
SCHEDULE=throughput ROUNDS=1 bash tools/step_rate.sh build/ab/lib_r03.so build/ab/lib_a1.so build/ab/lib_a2.so 2>&1 | tee $O/step_rate_throughput.log
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d.get('configs',{})
print('$1: %.0f solves/s %.2f ms chip %.3f success %.4f | one-at-a-time %.0f | lat3 %.0f | pipelined %.0f | tracking %.0f ref-params %.0f |' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['success_rate'], d['one_batch_at_a_time']['value'], d['latency_schedule_three_in_flight']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value']), {k:(round(v['value']), round(v['roofline']['chip_level_frac'],3)) for k,v in c.items()})"; }
for lib in build/ab/lib_r03.so build/ab/lib_a2.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "$lib driver-cmd" | tee -a $O/bench_ab.log
done
for K in 12 16 24; do
  BIOIK_SOLVE_DENSE_HANDOVER=$K BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=build/ab/lib_a2.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "lib_a2 handover=$K driver-cmd" | tee -a $O/bench_ab.log
done
for K in 0 16; do
  BIOIK_SOLVE_DENSE_HANDOVER=$K BIOIK_BENCH_CONFIGS=0 BIOIK_HIP_LIBRARY=build/ab/lib_a2.so python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib_a2 handover=$K 60 steps timed-only: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))" | tee -a $O/bench_ab.log
done
