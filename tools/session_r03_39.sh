#!/bin/bash
# round 3, GPU session 39: the throughput schedule on the four-wavefront dense kernel: bench line, solves in flight, host pipeline, GPU suite, parity soak
O=gpurun_out/s39; mkdir -p $O
export TMPDIR=/tmp
{
for nf in 5 6 8 12; do
  python bench.py --no-cpu-baseline --timed-only --in-flight $nf --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$nf in flight: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done
BIOIK_SOLVE_THREE_WAVES=1 python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('three wavefronts per SIMD, 6 in flight: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
python bench.py --no-cpu-baseline --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench: %.0f solves/s %.2f ms chip %.3f | latency schedule, three in flight %.0f | one-at-a-time %.0f | pipelined %.0f host %.0f | tracking %.0f ref-params %.0f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac'], d['latency_schedule_three_in_flight']['value'], d['one_batch_at_a_time']['value'], d['host_pointer_pipelined']['value'], d['host_pointer_entry']['solves_per_s'], d['tracking_seeds']['value'], d['reference_parameters']['value']))"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver command: %.0f solves/s %.2f ms' % (d['value'], d['ms_per_step']))"
python tools/pipeline_probe.py 2>&1 | grep -v Warning | tail -9
} 2>&1 | tee $O/w4.log
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -4 | tee $O/gputests.log
( time timeout 900 python tools/fuzz_parity.py 1000 29 ) > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
