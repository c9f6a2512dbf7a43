#!/bin/bash
# round 6: A/B of library builds on the driver's command (20 timed steps) -- headline value, chip fraction, C3 / C4 legs; tools/ab_r6.sh out_dir lib1.so lib2.so ...
O=$1; shift; mkdir -p $O
for lib in "$@"; do
  n=$(basename $lib .so)
  BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_SMALL=0 BIOIK_BENCH_STREAM=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d.get("configs", {})
print("%-24s value %.4g  frac %.3f  ms/step %.3f  success %.4f  mean steps %.2f | c3 %s  c4 %s" % (sys.argv[2], d["value"], d["roofline"]["frac"], d["ms_per_step"], d.get("success_rate", 0), d.get("mean_steps_per_solve", 0),
      c.get("c3", {}).get("value"), c.get("c4", {}).get("value")))
PY
done
