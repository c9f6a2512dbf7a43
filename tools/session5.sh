#!/bin/bash
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/s5_pytest.log 2>&1; echo "pytest rc=$?" >> $O/s5_pytest.log
tail -3 $O/s5_pytest.log
bash tools/profile_round.sh r02 > $O/s5_profile_round.log 2>&1
tail -5 $O/s5_profile_round.log
python -c "
import json; d=json.load(open('$O/bench_r02.json'))
print('bench', d['value'], d['ms_per_step'], d['success_rate'], 'one-at-a-time', d['one_batch_at_a_time']['value'], 'roofline', d['roofline']['frac'], d['roofline']['chip_level_frac'])
print('configs', {k:(v['value'],v['success_rate'],v['ms_per_step'],v['roofline']['frac']) for k,v in d.get('configs',{}).items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('query_parallel'), d['speedup_vs_cpu_1thread'])
"
BIOIK_BENCH_C5_BATCH=16384 python bench.py --config c5 --steps 3 --warmup 1 > $O/s5_c5.json 2> $O/s5_c5.err; cat $O/s5_c5.json | head -c 1500
