#!/bin/bash
# A/B session: GPU parity suite on the current library, then bench + fixed-work step rate of the variants under ab/
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/s9_pytest.log 2>&1; echo "pytest rc=$?" >> $O/s9_pytest.log
tail -3 $O/s9_pytest.log
bash tools/ab.sh ab/*.so 2>&1 | tee $O/s9_ab.log
bash tools/step_rate.sh ab/*.so 2>&1 | tee $O/s9_step_rate.log
