#!/bin/bash
# A/B of library builds: fixed-work step rate, then the bench line (three launches in flight) with its C3 / C4 sub-records.  usage: tools/ab_full.sh lib...
bash tools/step_rate.sh "$@"
for lib in "$@"; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib bench: %.0f solves/s %.2f ms success %.4f one-at-a-time %.0f | configs' % (d['value'], d['ms_per_step'], d['success_rate'], d['one_batch_at_a_time']['value']), {k:(round(v['value']),v['success_rate']) for k,v in d.get('configs',{}).items()})"
done
