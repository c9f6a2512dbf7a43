#!/bin/bash
# round 5, GPU session 32 (31 again, after the bound on the bisection loop): the profile of record on the final kernels (pre-selection by selection, shared secondary sums, one running sum of goals): GPU suite, smoke(), a parity soak, then tools/profile_round.sh r05 (bench line, rocprofv3 kernel
# statistics of the same command, PMC passes) and the driver's own bench command
O=gpurun_out/r05s32; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.log
( time python tools/fuzz_parity.py 3000 2468 ) 2>&1 | grep -v " ok$" | tail -6 | tee $O/fuzz.log
( time bash tools/profile_round.sh r05 ) > $O/profile_round.log 2>&1
tail -3 gpurun_out/bench_r05.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
tail -c 800 $O/bench_driver_cmd.json
