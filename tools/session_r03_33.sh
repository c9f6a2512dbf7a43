#!/bin/bash
# round 3, GPU session 33: the dense kernel of the throughput schedule -- two-minima top-2 for its 32-lane groups (z1), winner copy skipped when the elites stay (z2),
# both (z3) -- against the build before (z0): fixed work under the dense mapping and the bench line (six in flight)
O=gpurun_out/s33; mkdir -p $O
export TMPDIR=/tmp
{
for rep in 1 2; do for lib in build/lib_z0.so build/lib_z1.so build/lib_z2.so build/lib_z3.so; do
  v=$(BIOIK_BENCH_SCHEDULE=throughput BIOIK_BENCH_IN_FLIGHT=1 BIOIK_BENCH_STREAM=0 BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_DTWIST=1e-300 BIOIK_BENCH_MAX_STEPS=32 BIOIK_BENCH_BATCH=3072 python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms -> %.0f steps/ms chip-wide' % (d['ms_per_step'], 3072*32/d['ms_per_step']))")
  echo "$lib fixed work, dense mapping: $v"
done; done
for rep in 1 2 3; do for lib in build/lib_z0.so build/lib_z1.so build/lib_z2.so build/lib_z3.so; do
  BIOIK_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --timed-only --steps 60 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib bench: %.0f solves/s %.2f ms chip %.3f' % (d['value'], d['ms_per_step'], d['roofline']['chip_level_frac']))"
done; done
} 2>&1 | tee $O/dense_ab.log
