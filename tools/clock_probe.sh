#!/bin/bash
# Shader clock under load: every workgroup of the profiling build (-DBIOIK_PHASE_TIMING) records its shader-clock cycles
# (s_memtime, summed over the phases) and its start / end on the constant 100 MHz clock (s_memrealtime); their ratio is the
# clock the CU actually ran at.  usage: tools/clock_probe.sh build/libphase.so
lib=$1
for cfg in "1 64" "256 64" "4096 64" "16384 64"; do set -- $cfg
  BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=$2 BIOIK_BENCH_BATCH=$1 BIOIK_PHASE_DUMP=/tmp/phase.bin BIOIK_HIP_LIBRARY=$lib python bench.py --timed-only --no-cpu-baseline --in-flight 1 --steps 1 --warmup 0 > /tmp/tl.json 2>/dev/null
  python - <<PY
import numpy as np
a=np.fromfile("/tmp/phase.bin",dtype=np.uint64).reshape(-1,28)
ph=a[:,:24].astype(np.float64); ph[:,19:21]=0; cyc=ph.sum(axis=1); st=a[:,24].astype(np.float64); en=a[:,25].astype(np.float64)
us=(en-st)/100.0
f=cyc/us
print("batch %6d (every query runs $2 steps): workgroup time %.0f us mean, %.1f us/step; shader clock MHz: mean %.0f  p10 %.0f  p90 %.0f; launch span %.2f ms" % (len(us), us.mean(), us.mean()/$2, f.mean(), np.percentile(f,10), np.percentile(f,90), (en.max()-st.min())/1e5))
PY
done
