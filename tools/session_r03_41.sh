#!/bin/bash
# round 3, GPU session 41: the four-wavefront dense kernel with the species record read per generation (SLIM: 66 -> 30 spilled values): bench line, HBM traffic
O=gpurun_out/s41; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
{
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slim w4: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
  BIOIK_SOLVE_THREE_WAVES=1 python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w3: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
python bench.py --no-cpu-baseline --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench: %.0f solves/s | tracking %.0f | pipelined %.0f' % (d['value'], d['tracking_seeds']['value'], d['host_pointer_pipelined']['value']))"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  BIOIK_BENCH_STREAM=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$c -o $c -- python $R/bench.py --timed-only --steps 5 --warmup 1 --in-flight 1 > $R/$O/pmc_$c.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$R/$O/pmc_$c/**/*counter_collection.csv", recursive=True)[0]
tot=0; n=0
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("k_solve") and r["Counter_Name"]=="$c": tot+=float(r["Counter_Value"]); n+=1
print("$c per solve launch: %.1f KiB over %d dispatches" % (tot/max(n,1), n))
PY
done
} 2>&1 | tee $R/$O/slim.log
