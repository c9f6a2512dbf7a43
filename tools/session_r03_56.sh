#!/bin/bash
# round 3, GPU session 56: the driver's bench command three times with every stream opened before the warm-up (session 55 measured 6.9e5 with five of the
# ten streams first used inside the timed region), then the default command
O=gpurun_out/s56; mkdir -p $O
export TMPDIR=/tmp BIOIK_BENCH_CONFIGS=0
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('driver command: %.0f solves/s %.2f ms, %d in flight' % (d['value'], d['ms_per_step'], d['config']['batches_in_flight']))"
done 2>&1 | tee $O/driver_command.log
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('default command: %.0f solves/s %.2f ms, %d in flight, chip-level %.3f' % (d['value'], d['ms_per_step'], d['config']['batches_in_flight'], d['roofline']['chip_level_frac']))" 2>&1 | tee -a $O/driver_command.log
