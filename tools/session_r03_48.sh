#!/bin/bash
# the driver's own command with the automatic number of solves in flight (ten at K = 20)
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_command.json 2> gpurun_out/bench_driver_command.err
tail -c 600 gpurun_out/bench_driver_command.json
