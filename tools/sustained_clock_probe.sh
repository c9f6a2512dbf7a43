#!/bin/bash
# Shader clock and workgroup timing of the LAST launch of a sustained sequence (profiling build, -DBIOIK_PHASE_TIMING: the dump is
# rewritten by every launch).  usage: tools/sustained_clock_probe.sh build/libphase.so
lib=$1
for cfg in "1 1" "1 10" "1 40" "3 60"; do set -- $cfg
  BIOIK_PHASE_DUMP=/tmp/phase.bin BIOIK_HIP_LIBRARY=$lib python bench.py --timed-only --no-cpu-baseline --in-flight $1 --steps $2 --warmup 0 > /tmp/tl.json 2>/dev/null
  python - <<PY
import numpy as np, json
d=json.load(open("/tmp/tl.json"))
a=np.fromfile("/tmp/phase.bin",dtype=np.uint64).reshape(-1,28)
ph=a[:,:24].astype(np.float64); steps=ph[:,20].copy(); ph[:,19:21]=0; cyc=ph.sum(axis=1); st=a[:,24].astype(np.float64); en=a[:,25].astype(np.float64)
us=(en-st)/100.0
print("in flight $1, $2 launches: bench %.2f ms/batch | last launch: span %.2f ms, last start %.2f ms, workgroup mean %.0f us, %.1f us per step, shader clock %.0f MHz (p10 %.0f)" % (
    d["ms_per_step"], (en.max()-st.min())/1e5, (st.max()-st.min())/1e5, us.mean(), us.sum()/max(steps.sum(),1), (cyc/us).mean(), np.percentile(cyc/us,10)))
PY
done
