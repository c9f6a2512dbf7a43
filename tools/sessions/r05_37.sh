#!/bin/bash
# round 5, GPU session 37 (the round's last GPU seconds): the line search takes a candidate with a NaN gene for no candidate (quirk Q5) -- the HBM traffic counters of a solve once
# more on these kernel sources (FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh; profiles/traffic.json carries the sources' hash), then the GPU suite
R=$(pwd); O=$R/gpurun_out; mkdir -p $O/r05s37; export TMPDIR=/tmp
cd /tmp
pmc() { d=$1; shift; timeout 45 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$d -o $d -- python $R/bench.py --timed-only --steps 5 --warmup 1 --in-flight 1 > $O/pmc_$d.log 2>&1; }
BIOIK_BENCH_STREAM=0 pmc fetch FETCH_SIZE
BIOIK_BENCH_STREAM=0 pmc write WRITE_SIZE
cd $R
( time timeout 75 python -m pytest tests -m gpu -q ) > $O/r05s37/gpu_suite.log 2>&1; grep -E "passed|failed" $O/r05s37/gpu_suite.log
ls $O/pmc_fetch $O/pmc_write | head
