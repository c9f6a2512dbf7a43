#!/bin/bash
# round 5, GPU session 8: as session 7 after its two fixes (bench import, the plugin test compares calls of one size)
O=gpurun_out/r05s08; mkdir -p $O
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2>&1 | grep real
tail -c 900 $O/bench_driver_cmd.json; echo; tail -3 $O/bench_driver_cmd.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05s08/bench_driver_cmd.json') if l.startswith('{')][-1])
for e in d['small_batches']['sizes']: print(e)
PY
