#!/bin/bash
# GPU session: parity suite, then A/B of library builds (fixed work + bench workload), phase profile
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/s1_pytest.log 2>&1; echo "pytest rc=$?" >> $O/s1_pytest.log
tail -3 $O/s1_pytest.log
bash tools/step_rate.sh build/lib_r1.so bio_ik_amd/libbioik_hip.so > $O/s1_step_rate.log 2>&1
cat $O/s1_step_rate.log
bash tools/ab.sh build/lib_r1.so bio_ik_amd/libbioik_hip.so > $O/s1_ab.log 2>&1
cat $O/s1_ab.log
bash tools/lone_probe.sh build/libphase_new.so > $O/s1_phases_new.log 2>&1
python bench.py --no-cpu-baseline --steps 30 > $O/s1_bench.json 2> $O/s1_bench.err
python -c "
import json; d=json.load(open('$O/s1_bench.json'))
print('bench', d['value'], d['ms_per_step'], d['success_rate'], d['mean_steps_per_solve'], 'one-at-a-time', d['one_batch_at_a_time']['value'])"
