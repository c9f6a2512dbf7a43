#!/bin/bash
# round 5, GPU session 24: the final tree (ABI 5): GPU suite, smoke(), the two-rank dry run of `bench.py --gpus 2` on a one-GPU box (control flow of N > 1, gloo), C5 in miniature
mkdir -p gpurun_out; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > gpurun_out/r05s24_gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/r05s24_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r05s24_bench_gpus2_one_gpu_box.json 2> gpurun_out/r05s24_bench_gpus2.err ) 2>&1 | grep real; tail -c 600 gpurun_out/r05s24_bench_gpus2_one_gpu_box.json; echo
( time BIOIK_BENCH_C5_BATCH=16384 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05s24_bench_c5_16384.json 2>/dev/null ) 2>&1 | grep real; tail -c 400 gpurun_out/r05s24_bench_c5_16384.json; echo
