#!/bin/bash
# round 3, GPU session 35: the bench line of record (default command) after the last harness change
O=gpurun_out/s35; mkdir -p $O
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_r03.json 2> $O/bench.err
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
"
