#!/bin/bash
# round 4, GPU session 3: GPU suite on the tree with island_sync, the hybrid callback-goal path, bioik_eval_arith; the full default bench line (new CPU legs,
# oracle pose check); `python bench.py --gpus 2` by itself on a one-GPU box (re-executes under torch.distributed.run, gloo since ranks share the device)
O=gpurun_out/r04s3; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1
tail -5 $O/gpu_suite.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04s3/bench_default.json') if l.startswith('{')][-1])
print('bench: %.0f solves/s %.2f ms chip %.3f | lat3 %.0f | one-at-a-time %.0f | pipelined %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['reference_parameters']['value']))
cb=d['cpu_baseline']; print('cpu 1 thread', cb['value'], 'query_parallel', cb.get('query_parallel'), 'v3', cb.get('march_x86_64_v3'))
print('pose check:', d['pose_check'], d['max_pos_err_m_of_successes'], d['max_rot_err_rad_of_successes'])
print({k:(round(v['value']), round(v['roofline']['chip_level_frac'],3)) for k,v in d['configs'].items()})
PY
( time python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gpus2_plain.json 2> $O/bench_gpus2_plain.err ) 2>&1 | grep real
tail -2 $O/bench_gpus2_plain.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04s3/bench_gpus2_plain.json') if l.startswith('{')][-1])
print('--gpus 2 plain: n_gpus', d['n_gpus'], 'value %.0f' % d['value'], d['process_group'])
PY
