#!/bin/bash
# round 5, GPU session 13: k_solve_lean_cl4h (k_solve_lean_cl4 with two helper wavefronts) -- first light under a timeout, the lone step, small batches, an isolated 4096-query call
O=gpurun_out/r05s13; mkdir -p $O
export TMPDIR=/tmp
{
timeout 120 python tools/lone_call_overhead.py 1 2>&1 | grep "islands [18]:" || echo "lone_call_overhead: timeout or failure"
BIOIK_SOLVE_HELPED=0 timeout 120 python tools/lone_call_overhead.py 1 2>&1 | grep "islands [18]:"
SMALL_SIZES=1,16,64,256,512,1024 timeout 300 python tools/small_batches.py "cl4:BIOIK_SOLVE_HELPED=0;islands=-16" "helped1024:BIOIK_SOLVE_HELPED=1024;islands=-16" "helped2048:BIOIK_SOLVE_HELPED=2048;islands=-16" "helped1024_1island:BIOIK_SOLVE_HELPED=1024" "cl4_1island:BIOIK_SOLVE_HELPED=0" 2>&1 | grep -v amdgpu
for h in 0 1024; do echo "isolated 4096-query calls, BIOIK_SOLVE_HELPED=$h"; BIOIK_SOLVE_HELPED=$h timeout 200 python bench.py --timed-only --in-flight 1 --schedule latency --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f solves/s  %.3f ms per call' % (d['value'], d['ms_per_step']))"; done
} 2>&1 | tee $O/helped_kernel.log
