#!/bin/bash
# round 3, GPU session 47: solves in flight at the driver's command (--steps 20) and at the default (--steps 60) on the final kernels
O=gpurun_out/s47; mkdir -p $O
export TMPDIR=/tmp
{
for st in 20 60; do for nf in 4 5 6 10 12; do for rep in 1 2; do
  python bench.py --gpus 1 --steps $st --warmup 5 --no-cpu-baseline --timed-only --in-flight $nf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $st, $nf in flight: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done; done; done
} 2>&1 | tee $O/inflight.log
