#!/bin/bash
# round 5, GPU session 30: a tip's goals and the gene-only goals continue ONE running sum (tip_goals / nonlink_primary take the sum so far): GPU suite with the new
# goal-set cases, parity soak, A/B of the timed region (60 steps, three alternating rounds) and of the bench line against the previous commit's library
mkdir -p gpurun_out/r05s30; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > gpurun_out/r05s30/gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/r05s30/gpu_suite.log
( time timeout 600 python tools/fuzz_parity.py 3000 ) > gpurun_out/r05s30/fuzz_3000.log 2>&1; grep "cases," gpurun_out/r05s30/fuzz_3000.log
for round in 1 2 3; do for lib in build/ab/lib_r05_shared.so bio_ik_amd/libbioik_hip.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 3 2>/dev/null > gpurun_out/r05s30/timed_$(basename $lib .so)_$round.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r05s30/timed_$(basename $lib .so)_$round.json').read().strip().splitlines()[-1]); print('$lib timed-only 60 steps: %.0f solves/s %.3f ms' % (d['value'], d['ms_per_step']))"
done; done
for lib in build/ab/lib_r05_shared.so bio_ik_amd/libbioik_hip.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null > gpurun_out/r05s30/bench_$(basename $lib .so).json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r05s30/bench_$(basename $lib .so).json').read().strip().splitlines()[-1]); print('$lib bench: %.0f solves/s %.2f ms | configs' % (d['value'], d['ms_per_step']), {k:(round(v['value']),round(v['ms_per_step'],2),v['success_rate']) for k,v in d.get('configs',{}).items()})"
done
