#!/bin/bash
# instruction-mix counters of k_solve_lean on the fixed-work probe (3072 workgroups x 32 steps, no query may succeed)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $O/s3_counters_avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $O/s3_counters_avail.txt | sort -u > $O/s3_sq_names.txt
wc -l $O/s3_sq_names.txt
export BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=3072
pmc() { tag=$1; lib=$2; shift; shift; BIOIK_HIP_LIBRARY=$lib rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/s3_$tag -o $tag -- python $R/bench.py --timed-only --no-cpu-baseline --steps 3 --warmup 1 > $O/s3_$tag.log 2>&1; }
for v in r1:$R/build/lib_r1.so new:$R/bio_ik_amd/libbioik_hip.so; do
  n=${v%%:*}; lib=${v#*:}
  pmc ${n}_a $lib SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
  pmc ${n}_b $lib SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU
  pmc ${n}_c $lib SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pmc ${n}_d $lib SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_FLAT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM
done
cd $R
python - <<'PY'
import csv, glob, collections, os
O="gpurun_out"
for tag in sorted(glob.glob(O+"/s3_*_[abcd]")):
    for f in glob.glob(tag+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_solve" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(os.path.basename(tag), {k: "%.4g"%(sum(v)/len(v)) for k,v in agg.items()})
PY
