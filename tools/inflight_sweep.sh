#!/bin/bash
# Throughput of a stream of 4096-query batches against the number of launches kept in flight (bench.py --in-flight):
#   tools/inflight_sweep.sh [steps]      (run on the GPU box; prints one line per setting)
steps=${1:-24}
for rep in 1 2; do for n in 1 2 3 4 6; do
  v=$(python bench.py --timed-only --no-cpu-baseline --in-flight $n --steps $steps --warmup $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.2f ms/batch  kernel %.2f ms' % (d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms', float('nan'))))")
  echo "in flight $n : $v"
done; done
