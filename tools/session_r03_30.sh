#!/bin/bash
# round 3, GPU session 30: the bench line at the driver's command (--steps 20 --warmup 5) by solves in flight: end effects of a short timed region
O=gpurun_out/s30; mkdir -p $O
export TMPDIR=/tmp
{
for nf in 4 5 6 8 10; do for rep in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --timed-only --in-flight $nf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20, $nf in flight: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done; done
for rep in 1 2; do python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 60, 6 in flight: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"; done
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json ) 2>&1 | grep real
python -c "import json; d=json.load(open('$O/bench_driver_cmd.json')); print('driver command: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
} 2>&1 | tee $O/short_runs.log
