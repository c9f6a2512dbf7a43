#!/bin/bash
# round 3, GPU session 57 (= 55 after bench.py opens every stream before the warm-up): the profile of record again (bench line with six timed launches per stream in the C3 / C4 legs, kernel statistics, PMC passes)
# and the driver's own bench command
O=gpurun_out/s57; mkdir -p $O
export TMPDIR=/tmp
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), v['batches_timed'], round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python -c "
import json; d=json.loads([l for l in open('$O/bench_driver_cmd.json') if l.startswith('{')][-1]); c=d['configs']
print('driver command: %.0f solves/s %.2f ms, %d in flight; C3 %.0f (%.3f) C4 %.0f (%.3f)' % (d['value'], d['ms_per_step'], d['config']['batches_in_flight'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))"
