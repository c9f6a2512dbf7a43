#!/bin/bash
# round 3, GPU session 31: hardware queues beyond eight (bench line, host pipeline inside bench.py)
O=gpurun_out/s31; mkdir -p $O
export TMPDIR=/tmp
{
for q in 8 12 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q: %.0f solves/s %.2f ms | pipelined from host arrays %.0f | tracking %.0f | c3 %.0f c4 %.0f' % (d['value'], d['ms_per_step'], d['host_pointer_pipelined']['value'], d['tracking_seeds']['value'], d['configs']['c3']['value'], d['configs']['c4']['value']))"
done
} 2>&1 | tee $O/queues.log
