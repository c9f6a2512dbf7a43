#!/bin/bash
# round 4, GPU session 41: the final tree (persistent scratch, one-launch mapping on capturing streams): GPU suite, smoke(), a parity soak, then the profile of record (bench line, rocprofv3 kernel statistics of the same command,
# PMC passes) and the driver's own bench command
O=gpurun_out/r04s41; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.log
( time python tools/fuzz_parity.py 2000 1357 ) 2>&1 | grep -v " ok$" | tail -6 | tee $O/fuzz.log
( time bash tools/profile_round.sh r04 ) > $O/profile_round.log 2>&1
tail -3 gpurun_out/bench_r04.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python - <<'PY'
import json
for f in ('gpurun_out/bench_r04.json', 'gpurun_out/r04s41/bench_driver_cmd.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['configs']
    print('%s: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f | ref-params %.0f | cpu %.0f -> %.0fx | C3 %.0f (%.3f) C4 %.0f (%.3f)' % (f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value'], d['cpu_baseline']['value'], d['speedup_vs_cpu_1thread'], c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))
PY
