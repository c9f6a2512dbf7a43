#!/bin/bash
# schedule of k_solve's workgroups on the chip (profiling build, -DBIOIK_PHASE_TIMING): start/end of every workgroup on the
# 100 MHz wall clock, the CU it ran on, and the phase breakdown.  usage: tools/timeline_probe.sh build/libphase.so
lib=$1
for cfg in "4096 128" "4096 256" "16384 128"; do set -- $cfg
  BIOIK_BENCH_BATCH=$1 BIOIK_PHASE_DUMP=/tmp/phase.bin BIOIK_HIP_LIBRARY=$lib BIOIK_SOLVE_THREADS=$2 python bench.py --timed-only --in-flight 1 --no-cpu-baseline --steps 1 --warmup 0 > /tmp/tl.json 2>/dev/null
  python - <<PY
import numpy as np, json
a=np.fromfile("/tmp/phase.bin",dtype=np.uint64).reshape(-1,28)
ph=a[:,:8].astype(np.float64); st=a[:,24].astype(np.float64); en=a[:,25].astype(np.float64); hw=a[:,26]
t0=st.min(); st=(st-t0)/100.0; en=(en-t0)/100.0   # microseconds
names=["init","reproduce","fitness","selection","memetics","species","check","preselect"]
print("== batch $1 threads $2: kernel span %.2f ms; phases:" % (en.max()/1e3), ", ".join("%s %.1f%%"%(n,100*ph[:,i].sum()/ph.sum()) for i,n in enumerate(names)))
dur=en-st
print("   wg duration us: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f ; last start at %.2f ms" % (dur.mean(), np.median(dur), np.percentile(dur,90), np.percentile(dur,99), dur.max(), st.max()/1e3))
# residency over time
T=np.linspace(0,en.max(),41)
res=[int(((st<=t)&(en>t)).sum()) for t in T]
print("   resident workgroups at 2.5%% time steps:", res)
hwid=(hw & 0xffffffff); xcc=(hw>>32)&0xf
cu=((hwid>>8)&0xf) | (((hwid>>13)&0x7)<<4) | (xcc<<8)   # CU_ID, SE_ID, XCC
print("   distinct (xcc,se,cu) seen: %d" % len(np.unique(cu)))
# per-step time as a function of residency: duration / steps is not available here; report total wg-time
print("   sum of workgroup residency %.1f wg-ms -> mean residency %.0f wgs" % (dur.sum()/1e3, dur.sum()/en.max()))
PY
done
