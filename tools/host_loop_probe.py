"""A loop of one-pose calls through the host-pointer entry (what a MoveIt control loop does): per-call wall times, calls above 3 ms listed.
usage: host_loop_probe.py [same|distinct|reversed] [calls]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(t, device=0)
mode = sys.argv[1] if len(sys.argv) > 1 else "same"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
p = abi.default_solve_params(population=128, max_steps=64, random_seed=1, islands=abi.ISLANDS_AUTO)
sets = [make_queries(t, h.active_variables, h.fk_genes, 1, seed=2000 + r)[:2] for r in range(40)]
if mode == "reversed":
    sets = sets[::-1]
ts, st = [], []
for i in range(calls):
    s_, p_ = sets[0] if mode == "same" else sets[i % len(sets)]
    t0 = time.perf_counter(); r = h.solve_batch(p, s_, p_); ts.append(time.perf_counter() - t0); st.append(int(r[3][0]))
ts = np.array(ts) * 1e3
print(mode, "median %.3f ms; calls above 3 ms (index, ms, steps):" % np.median(ts), [(int(i), round(float(ts[i]), 1), st[i]) for i in np.nonzero(ts > 3.0)[0]])
