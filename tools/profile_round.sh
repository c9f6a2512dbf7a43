#!/bin/bash
# One GPU session = everything the round's profiles/ need: the bench line, rocprofv3 kernel stats of the same command, and the
# PMC passes (each counter group in its own run, with --kernel-trace only).  usage (on the GPU box, repo root): tools/profile_round.sh r01
rnd=${1:-r02}
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench_$rnd.json 2> $O/bench_$rnd.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$rnd -o $rnd -- python $R/bench.py --timed-only > $O/prof_bench.log 2>&1
pmc() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$d -o $d -- python $R/bench.py --timed-only --steps 5 --warmup 1 --in-flight 1 > $O/pmc_$d.log 2>&1; }
BIOIK_BENCH_STREAM=0 pmc fetch FETCH_SIZE
BIOIK_BENCH_STREAM=0 pmc write WRITE_SIZE
BIOIK_BENCH_STREAM=0 pmc sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY
BIOIK_BENCH_STREAM=0 pmc mem SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SALU
BIOIK_BENCH_STREAM=0 pmc ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_ACTIVE_INST_ANY
# the streamed-fitness kernel (population genotype array resident in HBM): its measured traffic next to its algorithmic bytes
spmc() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$d -o $d -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_$d.log 2>&1; }
spmc sfetch FETCH_SIZE
spmc swrite WRITE_SIZE
cd $R
ls $O/prof_$rnd $O/pmc_* | head -40
