#!/bin/bash
# round 3, GPU session 61 (run on two boxes): the default bench command and the driver's, final bench.py
mkdir -p gpurun_out/s61; export TMPDIR=/tmp
for a in "" "--gpus 1 --steps 20 --warmup 5"; do
python bench.py $a 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['configs']
print('bench.py $a: %.0f solves/s %.2f ms chip-level %.3f, %d in flight | latency schedule three in flight %.0f | one at a time %.0f | host-pointer pipeline %.0f | tracking %.0f | C3 %.0f (%.3f) C4 %.0f (%.3f) | cpu %.0f -> %.0fx' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['config']['batches_in_flight'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac'], d['cpu_baseline']['value'], d['speedup_vs_cpu_1thread']))"
done | tee -a gpurun_out/s61/bench_lines.log
