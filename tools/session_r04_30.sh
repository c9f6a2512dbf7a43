#!/bin/bash
# round 4, GPU session 30: the bench line with 32 hardware queues for its 20 + 6 streams (24 left the host-pointer leg and C3 / C4 sharing queues), 24 beside it
O=gpurun_out/r04s30; mkdir -p $O
export TMPDIR=/tmp
for q in 32 24 32; do
GPU_MAX_HW_QUEUES=$q python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('queues $q: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/queues.log
done
