#!/bin/bash
# round 4, GPU session 11: lib_a8 = AvoidJointLimitsGoal's free zone in C4's pre-selection (genes that no child of a generation can take out of it are not generated);
# C4 on the bench line against lib_a7, per-phase cycles, a 3000-case parity soak against the oracle
O=gpurun_out/r04s11; mkdir -p $O
for lib in build/ab/lib_a7.so build/ab/lib_a8.so; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('$lib: value %.0f | C3 %.0f (%.3f, success %.3f) | C4 %.0f (%.3f, success %.4f, %.1f ms)' % (d['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c3']['success_rate'], c['c4']['value'], c['c4']['roofline']['chip_level_frac'], c['c4']['success_rate'], c['c4']['ms_per_step']))" | tee -a $O/bench_ab.log
done
( time python tools/fuzz_parity.py 3000 8844 ) 2>&1 | grep -v " ok$" | tail -6 | tee $O/fuzz.log
