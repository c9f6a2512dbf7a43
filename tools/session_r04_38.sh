#!/bin/bash
# round 4, GPU session 38: the two best children of a generation from keys in the half-wavefront kernels (dense, joint walk) against the merging butterfly (lib_b5)

O=gpurun_out/r04s38; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
for rep in 1 2; do for lib in build/ab/lib_b5.so bio_ik_amd/libbioik_hip.so; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('$lib: value %.0f (%.3f) | C3 %.0f (%.3f) | C4 %.0f (%.3f)' % (d['value'], d['roofline']['chip_level_frac'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/ab.log
done; done
