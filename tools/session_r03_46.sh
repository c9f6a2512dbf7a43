#!/bin/bash
# round 3, GPU session 46: GPU suite, bench line, profile passes, parity soak -- the final sources of the round
O=gpurun_out/s46; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
tail -3 gpurun_out/bench_r03.err
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so | tail -2
( time timeout 900 python tools/fuzz_parity.py 1500 43 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -6 | tee gpurun_out/s46/pipeline.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s46/bench_driver_cmd.json ) 2>&1 | grep real; python -c "import json; d=json.load(open(\"gpurun_out/s46/bench_driver_cmd.json\")); print(\"driver command: %.0f solves/s %.2f ms, %d in flight\" % (d[\"value\"], d[\"ms_per_step\"], d[\"config\"][\"batches_in_flight\"]))"
