#!/bin/bash
# a third parity soak on the final sources: 20000 cases, another seed
mkdir -p gpurun_out
( time timeout 1200 python tools/fuzz_parity.py 20000 90210 ) > gpurun_out/fuzz_long2.log 2>&1
tail -4 gpurun_out/fuzz_long2.log
