#!/bin/bash
# round 5, GPU session 29: does the line search's shared secondary sum cost the headline (C2) anything?  The dense kernel now compiles no secondary sum at all.
# The new GPU test, then three alternating rounds of the timed region alone (60 steps) for the previous commit's library and this one, then the C3 / C4 records once more
mkdir -p gpurun_out/r05s29; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -k "secondary_goals_of_every_kind or preselection or function_level or trajectory_bit_exact" ) > gpurun_out/r05s29/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r05s29/gpu_tests.log
for round in 1 2 3; do for lib in build/ab/lib_r05_select.so bio_ik_amd/libbioik_hip.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 3 2>/dev/null > gpurun_out/r05s29/timed_$(basename $lib .so)_$round.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r05s29/timed_$(basename $lib .so)_$round.json').read().strip().splitlines()[-1]); print('$lib timed-only 60 steps: %.0f solves/s %.3f ms' % (d['value'], d['ms_per_step']))"
done; done
BIOIK_HIP_LIBRARY=bio_ik_amd/libbioik_hip.so python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null > gpurun_out/r05s29/bench_new.json
python -c "import sys,json; d=json.loads(open('gpurun_out/r05s29/bench_new.json').read().strip().splitlines()[-1]); print('new bench: %.0f solves/s %.2f ms | configs' % (d['value'], d['ms_per_step']), {k:(round(v['value']),round(v['ms_per_step'],2),v['success_rate']) for k,v in d.get('configs',{}).items()})"
