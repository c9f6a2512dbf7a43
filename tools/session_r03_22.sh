#!/bin/bash
# round 3, GPU session 22: GPU suite, the bench line + rocprofv3 kernel stats + PMC passes (one solve at a time) on the final kernels, parity soak
O=gpurun_out/s22; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json'))
print('bench: %.0f solves/s %.2f ms chip_frac %.3f one-at-a-time %.0f pipelined %.0f host %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s']))
print('cpu', d['cpu_baseline']['value'], 'speedup', d['speedup_vs_cpu_1thread'])
print({k:(round(v['value']), round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3), round(v.get('speedup_vs_cpu_1thread',0))) for k,v in d['configs'].items()})
"
( time timeout 900 python tools/fuzz_parity.py 1500 11 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
