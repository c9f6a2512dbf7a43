#!/bin/bash
# round 3, GPU session 54: the C3 / C4 legs of bench.py time ONE launch per stream (all in flight at once: start of the first to the end of the last,
# no steady state).  The same legs with 1 / 2 / 3 / 6 launches per stream, six and ten streams.
O=gpurun_out/s54; mkdir -p $O
export TMPDIR=/tmp
for inf in 6 10; do for rounds in 1 2 3 6; do
  r=$(BIOIK_BENCH_CONFIG_ROUNDS=$rounds timeout 600 python bench.py --no-cpu-baseline --in-flight $inf --steps $((inf * 2)) --warmup 2 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['configs']
print('C3 %.0f solves/s (chip-level %.3f, %d launches)  C4 %.0f solves/s (chip-level %.3f)' % (c['c3']['value'], c['c3']['roofline']['chip_level_frac'], c['c3']['batches_timed'], c['c4']['value'], c['c4']['roofline']['chip_level_frac']))")
  echo "$inf in flight, $rounds per stream: $r"
done; done 2>&1 | tee $O/config_rounds.log
