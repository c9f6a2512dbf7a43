#!/bin/bash
# the two-launch solve on the bench workload: hand-over after K steps (0 = one launch), stream of batches and isolated call
for K in ${KS:-0 4 6 8 10 12 16 24}; do
  BIOIK_SOLVE_TWO_PHASE=$K python bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$K: %.0f solves/s %.2f ms per batch (three in flight) | one at a time %.0f solves/s %.2f ms | success %.4f | configs' % (d['value'], d['ms_per_step'], d['one_batch_at_a_time']['value'], d['one_batch_at_a_time']['ms_per_step'], d['success_rate']), {k:round(v['value']) for k,v in d.get('configs',{}).items()}, 'tracking seeds %.0f reference parameters %.0f' % (d['tracking_seeds']['value'], d['reference_parameters']['value']))"
done
python bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('auto: %.0f solves/s %.2f ms | one at a time %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step'], d['one_batch_at_a_time']['value'], d['one_batch_at_a_time']['ms_per_step']))"
