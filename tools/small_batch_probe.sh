#!/bin/bash
# launches that cannot fill the chip: the launcher's default lane mapping against a forced 128-lane mapping.  usage: tools/small_batch_probe.sh [lib]
lib=${1:-bio_ik_amd/libbioik_hip.so}
for b in 1 64 512 768 1024; do for thr in default 128; do
  if [ $thr = default ]; then unset BIOIK_SOLVE_THREADS; else export BIOIK_SOLVE_THREADS=$thr; fi
  v=$(BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_BATCH=$b python bench.py --timed-only --in-flight 1 --no-cpu-baseline --steps 6 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.3f ms per launch' % (d['value'], d['ms_per_step']))")
  echo "batch=$b lanes=$thr : $v"
done; done
