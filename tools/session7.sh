#!/bin/bash
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/s7_pytest.log 2>&1; echo "pytest rc=$?" >> $O/s7_pytest.log
tail -3 $O/s7_pytest.log
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so
python bench.py --no-cpu-baseline --steps 30 2>/dev/null > $O/s7_bench.json; python -c "
import json; d=json.load(open('$O/s7_bench.json'))
print('bench', d['value'], d['ms_per_step'], d['success_rate'], 'one-at-a-time', d['one_batch_at_a_time']['value'])
print('configs', {k:(v['value'],v['success_rate'],v['ms_per_step'],v['roofline']['chip_level_frac']) for k,v in d.get('configs',{}).items()})"
for cl in 0 1; do BIOIK_SOLVE_COLUMNLESS=$cl BIOIK_SOLVE_THREADS=128 python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('forced columnless=$cl threads=128: configs', {k:(v['value'],v['ms_per_step']) for k,v in d.get('configs',{}).items()}, 'c2', d['value'])"; done
