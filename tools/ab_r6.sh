#!/bin/bash
# round 6: A/B of library builds / switches on the driver's command (20 timed steps) -- headline value, chip fraction, C3 / C4 legs
# usage: tools/ab_r6.sh out_dir name=lib.so[,ENV=value ...] ...      (BIOIK_AB_STEPS=K: K timed steps instead of 20)
O=$1; shift; mkdir -p $O
for spec in "$@"; do
  n=${spec%%=*}; rest=${spec#*=}; lib=${rest%%,*}; envs=""
  if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
  env $envs BIOIK_HIP_LIBRARY=$lib BIOIK_BENCH_SMALL=0 BIOIK_BENCH_STREAM=0 python bench.py --gpus 1 --steps ${BIOIK_AB_STEPS:-20} --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json "$n" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("%-24s FAILED (%s)" % (sys.argv[2], e)); sys.exit(0)
c = d.get("configs", {})
s = d.get("summary", {})
print("%-24s value %.4g  frac %.3f  ms/step %.3f  success %.4f  mean steps %.2f | lat3 %s one %s | c3 %.4g  c4 %.4g" % (sys.argv[2], d["value"], d["roofline"]["frac"], d["ms_per_step"], d.get("success_rate", 0), d.get("mean_steps_per_solve", 0),
      s.get("lat3"), s.get("one_at_a_time"), c.get("c3", {}).get("value", 0), c.get("c4", {}).get("value", 0)))
PY
done
