#!/bin/bash
# round 5, GPU session 15: island counts around the helped kernel's 1024-unit limit; the drain threshold of an isolated 4096-query call with the helped kernel as its consumer
O=gpurun_out/r05s15; mkdir -p $O
export TMPDIR=/tmp
{
SMALL_SIZES=128,256,512 timeout 300 python tools/small_batches.py "islands2:;islands=2" "islands4:;islands=4" "islands8:;islands=8" 2>&1 | grep -v amdgpu
for d in 512 1024 1536 2048 3072; do echo -n "isolated 4096-query calls, BIOIK_SOLVE_DRAIN_BELOW=$d: "; BIOIK_SOLVE_DRAIN_BELOW=$d timeout 200 python bench.py --timed-only --in-flight 1 --schedule latency --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.3f ms per call' % (d['value'], d['ms_per_step']))"; done
for d in 1024 2048; do echo -n "three in flight, latency schedule, BIOIK_SOLVE_DRAIN_BELOW=$d: "; BIOIK_SOLVE_DRAIN_BELOW=$d timeout 200 python bench.py --timed-only --in-flight 3 --schedule latency --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.3f ms per call' % (d['value'], d['ms_per_step']))"; done
} 2>&1 | tee $O/helped_islands_drain.log
