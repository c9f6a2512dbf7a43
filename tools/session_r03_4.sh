#!/bin/bash
# round 3, GPU session 4: mapping (A) vs (B) micro-benchmark; solves in flight x hardware queues; where the measured HBM bytes of a solve come from
O=gpurun_out/s4; mkdir -p $O
export TMPDIR=/tmp
./build/mapping_b > $O/mapping_b.log 2>&1; cat $O/mapping_b.log
for q in 4 8; do for f in 3 4 5 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --timed-only --steps 36 --warmup 6 --in-flight $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hw queues $q in flight $f: %.0f solves/s %.2f ms per batch, per-solve kernels %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done 2>&1 | tee $O/inflight_hwq.log
cd /tmp
pmc() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$d -o $d -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 4 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/pmc_$d.log 2>&1; }
BIOIK_SOLVE_TWO_PHASE=0 pmc one_f FETCH_SIZE
BIOIK_SOLVE_TWO_PHASE=0 pmc one_w WRITE_SIZE
BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_MAX_STEPS=4 pmc one4_f FETCH_SIZE
BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_MAX_STEPS=4 pmc one4_w WRITE_SIZE
BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_BATCH=1024 pmc one1k_f FETCH_SIZE
BIOIK_SOLVE_TWO_PHASE=0 BIOIK_BENCH_BATCH=1024 pmc one1k_w WRITE_SIZE
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/s4/hbm_sources.log
import csv, glob, collections
for d in ("one_f","one_w","one4_f","one4_w","one1k_f","one1k_w"):
    for f in glob.glob("gpurun_out/s4/pmc_%s/*counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_solve" in r["Kernel_Name"]: agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items(): print(d, k, "launches %d mean %.0f KiB" % (len(v), sum(v) / len(v)))
PY
