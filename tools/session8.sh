#!/bin/bash
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/s8_pytest.log 2>&1; echo "pytest rc=$?" >> $O/s8_pytest.log
tail -3 $O/s8_pytest.log
BIOIK_SOLVE_REPORT=1 python bench.py --no-cpu-baseline --steps 10 2> $O/s8_report.err > $O/s8_bench.json; grep bioik $O/s8_report.err | sort | uniq -c | sort -rn | head -5
python -c "
import json; d=json.load(open('$O/s8_bench.json'))
print('auto: c2', d['value'], 'configs', {k:(v['value'],v['ms_per_step']) for k,v in d.get('configs',{}).items()})"
for cl in 1 2; do BIOIK_SOLVE_COLUMNLESS=$cl BIOIK_SOLVE_THREADS=128 python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('forced columnless=$cl threads=128: configs', {k:(v['value'],v['ms_per_step']) for k,v in d.get('configs',{}).items()}, 'c2', d['value'])"; done
