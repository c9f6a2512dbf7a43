#!/bin/bash
# round 4, GPU session 31: C3 / C4 on the bench line in ONE session: the library of commit 11d6e3b (before the register sort and the hand-over bookkeeping) against the
# present one, with ten streams in the process and with twenty (of which C3 / C4 use ten)
O=gpurun_out/r04s31; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for lib in build/ab/lib_clj4.so bio_ik_amd/libbioik_hip.so; do for inf in 10 20; do
BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 --in-flight $inf 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['configs']
print('$lib, $inf in flight: value %.0f (%.3f) | tracking %.0f | ref-params %.0f | C3 %.0f (%.3f) | C4 %.0f (%.3f)' % (d['value'], d['roofline']['chip_level_frac'], d['tracking_seeds']['value'], d['reference_parameters']['value'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))" | tee -a $O/ab.log
done; done; done
