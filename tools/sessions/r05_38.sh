#!/bin/bash
# round 5, GPU session 38: session 37's two HBM traffic passes once more (a comment of the kernel sources changed behind them, and profiles/traffic.json carries the sources' hash)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
pmc() { d=$1; shift; timeout 30 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$d -o $d -- python $R/bench.py --timed-only --steps 5 --warmup 1 --in-flight 1 > $O/pmc_$d.log 2>&1; }
BIOIK_BENCH_STREAM=0 pmc fetch FETCH_SIZE
BIOIK_BENCH_STREAM=0 pmc write WRITE_SIZE
ls $O/pmc_fetch $O/pmc_write | head
