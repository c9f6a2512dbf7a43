#!/bin/bash
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/s4_pytest.log 2>&1; echo "pytest rc=$?" >> $O/s4_pytest.log
tail -5 $O/s4_pytest.log
bash tools/step_rate.sh bio_ik_amd/libbioik_hip.so > $O/s4_step_rate.log 2>&1; cat $O/s4_step_rate.log
python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('bench', d['value'], d['ms_per_step'], d['success_rate'], d['mean_steps_per_solve'], 'one-at-a-time', d['one_batch_at_a_time']['value'])"
