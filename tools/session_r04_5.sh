#!/bin/bash
# round 4, GPU session 5: PMC passes of the dense kernel on FIXED work (no query may succeed, 32 steps, 4096 queries, throughput schedule, one launch at a time):
# instruction classes, VALU busy, waits -- what bounds k_solve_lean_cl64w4 after the register spills are gone
O=$(pwd)/gpurun_out/r04s5; mkdir -p $O
R=$(pwd)
export TMPDIR=/tmp
export BIOIK_BENCH_SCHEDULE=throughput BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=4096
python bench.py --no-cpu-baseline --timed-only --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fixed work, 4096 x 32 steps: %.3f ms per launch -> %.0f steps/ms' % (d['ms_per_step'], 4096*32/d['ms_per_step']))" | tee $O/fixed_work.log
cd /tmp
pmc() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$d -o $d -- python $R/bench.py --no-cpu-baseline --timed-only --steps 3 --warmup 1 > $O/pmc_$d.log 2>&1; }
pmc sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY
pmc mem SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SALU
pmc mix SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64
pmc busy SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R
python - <<'PY'
import csv, glob, collections, os
O='gpurun_out/r04s5'
for f in sorted(glob.glob(O+'/pmc_*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_solve' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
            disp={k:r[k] for k in ('Grid_Size','Workgroup_Size','LDS_Block_Size','Scratch_Size','VGPR_Count','SGPR_Count')}
    for (k,c),v in sorted(agg.items()):
        print('%-28s %-28s launches %d mean %.5g' % (k,c,len(v),sum(v)/len(v)))
    print(os.path.basename(f), disp)
PY
