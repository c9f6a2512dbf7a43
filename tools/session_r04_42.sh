#!/bin/bash
# round 4, GPU session 42: after the host-side change of the scratch (stream-ordered, persistent): the two GPU tests it touches, and the HBM-traffic passes again so that
# profiles/traffic.json carries the hash of the final sources (device code unchanged since session 41, whose other passes stay)
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pipelining or hipgraph or stragglers or islands" 2>&1 | tail -2
cd /tmp
pmc() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$d -o $d -- python $R/bench.py --timed-only --steps 5 --warmup 1 --in-flight 1 > $O/pmc_$d.log 2>&1; }
rm -rf $O/pmc_fetch $O/pmc_write
BIOIK_BENCH_STREAM=0 pmc fetch FETCH_SIZE
BIOIK_BENCH_STREAM=0 pmc write WRITE_SIZE
ls $O/pmc_fetch $O/pmc_write | head
