#!/bin/bash
# round 3, GPU session 58: the host-pointer pipeline leg of bench.py fell from 9.2e5 to 7.4e5 when the device-pointer legs went from six to ten streams
# (ten + six I/O streams + the default stream on 16 hardware queues).  GPU_MAX_HW_QUEUES 16 / 24 / 32, and the stand-alone probe.
O=gpurun_out/s58; mkdir -p $O
export TMPDIR=/tmp BIOIK_BENCH_CONFIGS=0
for q in 16 24 32; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('hardware queues %s: %.0f solves/s, %d in flight | host-pointer pipeline %.0f | host-pointer entry %.0f | tracking %.0f' % (d['config']['hardware_queues'], d['value'], d['config']['batches_in_flight'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value']))"
done 2>&1 | tee $O/hwq.log
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -6 | tee -a $O/hwq.log
