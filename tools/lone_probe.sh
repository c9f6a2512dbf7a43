#!/bin/bash
# critical path of ONE workgroup (nobody else on the chip) and of a full chip, per phase: profiling build (-DBIOIK_PHASE_TIMING),
# no query may succeed (dtwist 1e-300) so every workgroup runs exactly 32 steps.  usage: tools/lone_probe.sh build/libphase.so
lib=$1
for cfg in "1 128" "1 256" "1536 128"; do set -- $cfg
  BIOIK_BENCH_STREAM=0 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=$1 BIOIK_PHASE_DUMP=/tmp/phase.bin BIOIK_HIP_LIBRARY=$lib BIOIK_SOLVE_THREADS=$2 python bench.py --timed-only --in-flight 1 --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
  python - <<PY
import numpy as np
a=np.fromfile("/tmp/phase.bin",dtype=np.uint64).reshape(-1,28).astype(np.float64)
names=["init","reproduce","fitness","selection","memetics","species","check","preselect","sel.top2","sel.xwave","sel.copy","sel.barrier","mem.approx","mem.grad","mem.norm","mem.line","mem.accept","mem.tail","rank","#mem_iter","#steps","linearise","mem.support_cols","mem.support_eval"]
m=a.mean(axis=0); steps=m[20]; tot=m[:19].sum()+m[21]+m[22]+m[23]
wall=(a[:,25]-a[:,24]).mean()*10.0  # ns
print("== batch $1 threads $2: %.1f steps, %.1f us/step wall, %.0f shader cycles/step (lane 0 of the workgroup), %.2f memetic iterations per step (both species counted on wave 0 only)" % (steps, wall/1e3/steps, tot/steps, m[19]/steps))
for i,n in enumerate(names):
    if n.startswith("#") or m[i]==0: continue
    print("   %-12s %8.0f cycles/step  %5.1f%%  %6.2f us" % (n, m[i]/steps, 100*m[i]/tot, m[i]/steps/tot*steps*wall/1e3/steps))
PY
done
