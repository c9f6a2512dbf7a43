#!/bin/bash
# round 4, GPU session 39: keyed top-2 (present library) against the merging butterfly (build/ab/lib_b5.so) on the headline: 60 timed steps, ten in flight, alternating
O=gpurun_out/r04s39; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3 4; do for lib in build/ab/lib_b5.so bio_ik_amd/libbioik_hip.so; do
echo "$lib: $(BIOIK_HIP_LIBRARY=$lib python bench.py --timed-only --no-cpu-baseline --steps 60 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s %.3f ms chip %.4f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))")" | tee -a $O/ab.log
done; done
