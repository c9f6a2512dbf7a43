#!/bin/bash
# round 4, GPU session 15: the pre-selection's sort in registers (sort_pairs_in_registers) on C3 / C4 against session 13's numbers (C3 44.6e3, C4 72.8e3), GPU suite, phases
O=gpurun_out/r04s15; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('value %.0f | C3 %.0f (%.3f, success %.3f, %.1f ms) | C4 %.0f (%.3f, success %.4f, %.1f ms)' % (d['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c3']['success_rate'], c['c3']['ms_per_step'], c['c4']['value'], c['c4']['roofline']['chip_level_frac'], c['c4']['success_rate'], c['c4']['ms_per_step']))" | tee -a $O/bench_ab.log
done
export BIOIK_HIP_LIBRARY=build/ab/libphase.so BIOIK_SOLVE_REPORT=1
( python tools/phase_probe_config.py c3 3072; python tools/phase_probe_config.py c4 2048 ) 2>&1 | grep -v "amdgpu.ids" | tee $O/phases.log | grep -E "==|presel|fitness"
