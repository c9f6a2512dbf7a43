#!/bin/bash
# per-phase cycle breakdown of k_solve (profiling build)
lib=$1
for thr in 64 128 256; do
  BIOIK_PHASE_DUMP=/tmp/phase_$thr.bin BIOIK_HIP_LIBRARY=$lib BIOIK_SOLVE_THREADS=$thr python bench.py --timed-only --in-flight 1 --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2>&1
  python - <<PY
import numpy as np
a=np.fromfile("/tmp/phase_$thr.bin",dtype=np.uint64).reshape(-1,28)[:,:8].astype(np.float64)
names=["init","reproduce","fitness","selection","memetics","species","check","preselect"]
tot=a.sum()
print("threads=$thr total cycles/query %.0f :" % (tot/len(a)), ", ".join("%s %.1f%%"%(n,100*a[:,i].sum()/tot) for i,n in enumerate(names)))
PY
done
