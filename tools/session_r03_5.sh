#!/bin/bash
# round 3, GPU session 5: cooperative single-individual walks (COOP) A/B, GPU suite, bench, per-phase cycles, PMC traffic one solve at a time
O=gpurun_out/s5; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1
tail -3 $O/gputests.log
ROUNDS=2 bash tools/step_rate.sh build/lib_f_all.so build/lib_k_nocoop.so build/lib_k_coop.so > $O/step_rate.log 2>&1
cat $O/step_rate.log
for lib in build/lib_k_nocoop.so build/lib_k_coop.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null > $O/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.load(open('$O/bench_$(basename $lib .so).json')); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f chip_frac %.3f pipelined %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value'], d['roofline'].get('chip_level_frac', -1), d['host_pointer_pipelined']['value']), {k:(round(v['value']),round(v['success_rate'],3), round(v['roofline']['chip_level_frac'],3)) for k,v in d.get('configs',{}).items()})"
done 2>&1 | tee $O/bench_ab.log
for c in c2 c3 c4; do BIOIK_SOLVE_REPORT=1 BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py $c $([ $c = c2 ] && echo 1536 || echo 3072); done > $O/phases.log 2>&1
bash tools/lone_probe.sh build/libphase.so >> $O/phases.log 2>&1
grep -E "^==|fitness|reproduce|rank|approx|species|support_eval|top2" $O/phases.log
cd /tmp
pmc() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$d -o $d -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 4 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/pmc_$d.log 2>&1; }
pmc two_f FETCH_SIZE
pmc two_w WRITE_SIZE
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/s5/hbm_two_launch.log
import csv, glob, collections
for d in ("two_f","two_w"):
    for f in glob.glob("gpurun_out/s5/pmc_%s/*counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_solve" in r["Kernel_Name"]: agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items(): print(d, k, "launches %d mean %.0f KiB" % (len(v), sum(v) / len(v)))
PY
