#!/bin/bash
# kernel start / end times of a stream of batches (three in flight), one-launch and two-launch solves: rocprofv3 --kernel-trace
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for TP in 0 1; do
  BIOIK_SOLVE_TWO_PHASE=$TP rocprofv3 --kernel-trace --output-format csv -d $O/trace_tp$TP -o t -- python $R/bench.py --timed-only --no-cpu-baseline --steps 30 --warmup 6 > $O/trace_tp$TP.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob
for tp in (0, 1):
    f = glob.glob('gpurun_out/trace_tp%d/**/t_kernel_trace.csv' % tp, recursive=True) + glob.glob('gpurun_out/trace_tp%d/t_kernel_trace.csv' % tp)
    rows = [r for r in csv.DictReader(open(f[0])) if 'k_solve' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    t0 = int(rows[0]['Start_Timestamp'])
    print('== two_phase=%d: %d kernels' % (tp, len(rows)))
    for r in rows[18:48]:
        s, e = (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6
        print('%-16s stream %s start %8.2f end %8.2f dur %6.2f' % (r['Kernel_Name'].split('(')[0], r['Stream_Id'], s, e, e - s))
PY
