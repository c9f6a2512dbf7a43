#!/bin/bash
# round 4, GPU session 4: the parents' mixed momentum from a table.  lib_a3: computed per gene and child; lib_a4: six-row table of the finished momentum term
# (dense kernel only, +768 B of LDS per query); lib_a5: two-row table of the mixed momentum on the species' other elite buffer (no LDS growth), both fixed-mapping kernels
# incl. the pre-selection pass of C4
O=gpurun_out/r04s4; mkdir -p $O
export TMPDIR=/tmp
SCHEDULE=throughput ROUNDS=2 bash tools/step_rate.sh build/ab/lib_a3.so build/ab/lib_a4.so build/ab/lib_a5.so 2>&1 | tee $O/step_rate_throughput.log
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d.get('configs',{})
print('$1: %.0f solves/s %.2f ms chip %.3f success %.4f | one-at-a-time %.0f | lat3 %.0f | pipelined %.0f | tracking %.0f ref-params %.0f |' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['success_rate'], d['one_batch_at_a_time']['value'], d['latency_schedule_three_in_flight']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value']), {k:(round(v['value']), round(v['roofline']['chip_level_frac'],3)) for k,v in c.items()})"; }
for lib in build/ab/lib_a3.so build/ab/lib_a5.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line "$lib driver-cmd" | tee -a $O/bench_ab.log
done
BIOIK_HIP_LIBRARY=build/ab/lib_a5.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
