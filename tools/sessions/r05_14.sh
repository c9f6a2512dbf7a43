#!/bin/bash
# round 5, GPU session 14: k_solve_lean_cl4h with back-to-back first polls; the product's own island rule under BIOIK_SOLVE_HELPED = 0 / 1024 / 2048; GPU suite
O=gpurun_out/r05s14; mkdir -p $O
export TMPDIR=/tmp
{
timeout 120 python tools/lone_call_overhead.py 1 2>&1 | grep "islands [18]:"
SMALL_SIZES=1,16,64,128,256,512,1024 timeout 300 python tools/small_batches.py "cl4_auto:BIOIK_SOLVE_HELPED=0;islands=0" "helped1024_auto:BIOIK_SOLVE_HELPED=1024;islands=0" "helped2048_auto:BIOIK_SOLVE_HELPED=2048;islands=0" 2>&1 | grep -v amdgpu
for h in 0 1024; do echo "isolated 4096-query calls, BIOIK_SOLVE_HELPED=$h"; BIOIK_SOLVE_HELPED=$h timeout 200 python bench.py --timed-only --in-flight 1 --schedule latency --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.3f ms per call' % (d['value'], d['ms_per_step']))"; done
} 2>&1 | tee $O/helped_kernel.log
( time timeout 600 python -m pytest tests -m gpu -q -x ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed" $O/gpu_suite.log; grep -n "^E " $O/gpu_suite.log | head -5
