#!/bin/bash
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/s2_pytest.log 2>&1; echo "pytest rc=$?" >> $O/s2_pytest.log
tail -3 $O/s2_pytest.log
bash tools/step_rate.sh build/lib_r1.so bio_ik_amd/libbioik_hip.so build/lib_sc2.so > $O/s2_step_rate.log 2>&1
cat $O/s2_step_rate.log
bash tools/ab.sh build/lib_r1.so bio_ik_amd/libbioik_hip.so build/lib_sc2.so > $O/s2_ab.log 2>&1
cat $O/s2_ab.log
for lib in bio_ik_amd/libbioik_hip.so build/lib_sc2.so; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('bench $lib', d['value'], d['ms_per_step'], d['success_rate'], d['mean_steps_per_solve'], 'one-at-a-time', d['one_batch_at_a_time']['value'])"
done
python tools/steps_hist.py 256 > $O/s2_steps_hist.log 2>&1; cat $O/s2_steps_hist.log
