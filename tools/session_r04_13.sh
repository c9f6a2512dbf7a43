#!/bin/bash
# round 4, GPU session 13: k_solve_lean_clj4 (the joint walk of both species' children under 128 registers, 4 wavefronts per SIMD) on C3 against the
# three-wavefront kernel of the same mapping (BIOIK_SOLVE_THREE_WAVES), the GPU parity suite on this tree
O=gpurun_out/r04s13; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
for tw in 0 1 0 1; do
E=""; [ $tw = 1 ] && E="BIOIK_SOLVE_THREE_WAVES=1"
env $E python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('three_waves=$tw: value %.0f | C3 %.0f (%.3f, success %.3f, %.1f ms) | C4 %.0f (%.3f, success %.4f, %.1f ms)' % (d['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c3']['success_rate'], c['c3']['ms_per_step'], c['c4']['value'], c['c4']['roofline']['chip_level_frac'], c['c4']['success_rate'], c['c4']['ms_per_step']))" | tee -a $O/bench_ab.log
done
BIOIK_SOLVE_REPORT=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep 'launch:' | sort | uniq -c | sort -rn | head -12 | tee -a $O/bench_ab.log
