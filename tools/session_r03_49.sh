#!/bin/bash
# a longer parity soak on the final sources: 20000 further cases (another seed), whole solves bit for bit against the oracle
mkdir -p gpurun_out
( time timeout 1200 python tools/fuzz_parity.py 20000 4711 ) > gpurun_out/fuzz_long.log 2>&1
tail -4 gpurun_out/fuzz_long.log
