#!/bin/bash
# round 5, GPU session 28: the line search's secondary sums with shared terms (secondary_shared) and the sums over the joint values in gene order where the genes do not
# follow the ops: GPU suite, parity soak (two new fixtures with several secondary goals), A/B of the bench line with its C3 / C4 records against the library of the
# previous commit (build/ab/lib_r05_select.so), per-phase cycles of C3 / C4
mkdir -p gpurun_out/r05s28; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > gpurun_out/r05s28/gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/r05s28/gpu_suite.log
( time timeout 600 python tools/fuzz_parity.py 3000 ) > gpurun_out/r05s28/fuzz_3000.log 2>&1; grep "cases," gpurun_out/r05s28/fuzz_3000.log
for round in 1 2; do for lib in build/ab/lib_r05_select.so bio_ik_amd/libbioik_hip.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null > gpurun_out/r05s28/bench_$(basename $lib .so)_$round.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r05s28/bench_$(basename $lib .so)_$round.json').read().strip().splitlines()[-1]); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value']), {k:(round(v['value']),round(v['ms_per_step'],2),v['success_rate']) for k,v in d.get('configs',{}).items()})"
done; done
for cfg in c3 c4; do BIOIK_SOLVE_AUTOTUNE=0 BIOIK_SOLVE_REPORT=1 BIOIK_HIP_LIBRARY=build/libphase.so python tools/phase_probe_config.py $cfg $([ $cfg = c3 ] && echo 3072 || echo 2048) 2>&1 | grep -v "^\[bioik\] joint"; done > gpurun_out/r05s28/phases_c3_c4.log 2>&1
grep -E "^==|support" gpurun_out/r05s28/phases_c3_c4.log
