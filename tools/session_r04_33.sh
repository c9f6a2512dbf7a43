#!/bin/bash
# round 4, GPU session 34: the joint-walk kernel with the walk for serial segments (lib_b3: the general walk), same session

O=gpurun_out/r04s34; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
for rep in 1 2; do for lib in build/ab/lib_b3.so bio_ik_amd/libbioik_hip.so; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('$lib: value %.0f (%.3f) | lat3 %.0f | one-at-a-time %.0f | tracking %.0f | ref-params %.0f | C3 %.0f (%.3f) | C4 %.0f (%.3f)' % (d['value'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/ab.log
done; done
BIOIK_SOLVE_REPORT=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep 'solve:' | sort | uniq -c | sort -rn | head -8 | cut -c1-330 | tee -a $O/ab.log
