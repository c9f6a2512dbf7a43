"""Lanes-per-query / stored-children / paired-evaluation sweep for C3 and C4 (the launcher's defaults are tuned on C2)."""
import itertools
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import tools.config_sweep as cs
from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, pr2_like, snake
which = sys.argv[1]
if which == "c3":
    t = ProblemTemplate(pr2_like(), "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), cs.sec(MinimalDisplacementGoal())])
    cs.run("C3", t, 128, "global", 64, steps=4)
else:
    t = ProblemTemplate(snake(31), "snake", [PoseGoal("tip"), cs.sec(AvoidJointLimitsGoal())])
    cs.run("C4", t, 512, "global", 32, steps=4)
''' % ROOT
for which in ("c3", "c4"):
    for thr, store, pairs in itertools.product(("64", "128", "256"), ("0", "1"), ("0", "1")):
        if store == "0" and pairs == "1":
            continue
        env = dict(os.environ, BIOIK_SOLVE_THREADS=thr, BIOIK_SOLVE_CHILD_PAIRS=pairs)
        if store == "0":
            env["BIOIK_SOLVE_STORE_CHILDREN"] = "0"
        r = subprocess.run([sys.executable, "-c", code, which], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("C")]
        print("threads=%s store=%s pairs=%s : %s" % (thr, store, pairs, line[0][40:] if line else r.stderr[-200:]), flush=True)
