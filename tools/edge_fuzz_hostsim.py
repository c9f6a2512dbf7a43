"""Edge-parameter soak WITHOUT a GPU: the kernel bodies in the host simulator (tests/hostsim) against the CPU oracle, whole solves bit
for bit, on the corners the regular suites visit only singly — one-joint and two-joint chains, populations of 1 / 2 / 3 / 65 children,
one and three islands, both modes, exact and linearised phenotypes, secondary goals at the smallest population that admits them.
usage: python tools/edge_fuzz_hostsim.py        (a few minutes of CPU time; prints one line per case, exit code 1 on a mismatch)"""
import itertools
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hostsim"), "-s"], check=True)
from bio_ik_amd import solver, abi, ProblemTemplate, PoseGoal, PositionGoal, AvoidJointLimitsGoal, MinimalDisplacementGoal, pr2_like, snake
from bio_ik_amd.workload import make_queries
from oracle import orc
lib = solver.load_library(os.path.join(ROOT, 'tests', 'hostsim', 'libbioik_hostsim.so'))
orc.set_trig_mode(1)
T = {"snake1": ProblemTemplate(snake(1), "snake", [PoseGoal("tip")]),
     "snake2+sec": ProblemTemplate(snake(2), "snake", [PositionGoal("tip"), AvoidJointLimitsGoal()]),
     "snake9": ProblemTemplate(snake(9), "snake", [PoseGoal("tip"), MinimalDisplacementGoal()]),
     "c2": ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])}
bad = 0; n = 0
for name, t in T.items():
    h, o = solver.HipSolver(t, lib=lib), orc.Oracle(t)
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 2, seed=3)
    for pop, islands, steps, mode, fk in itertools.product((1, 2, 3, 65), (1, 3), (1,), ("bio2_memetic", "bio2"), (abi.FK_EXACT, abi.FK_LINEAR)):
        if "sec" in name and pop < 2: continue
        if name == "snake9" and pop < 2: continue
        p = abi.default_solve_params(mode=mode, population=pop, max_steps=steps, random_seed=7, fk_mode=fk, islands=islands)
        try:
            got = h.solve_batch(p, seeds, params)
        except Exception as e:
            print("ERR", name, pop, islands, steps, mode, fk, e); bad += 1; continue
        want = o.solve_batch(p, orc.RNG_COUNTER, seeds, params)
        ok = all(np.array_equal(a, b) for a, b in zip(got, want))
        n += 1; print("ok" if ok else "BAD", name, pop, islands, mode, fk, flush=True)
        if not ok:
            bad += 1; print("MISMATCH", name, pop, islands, steps, mode, fk)
print(n, "cases", bad, "bad")
sys.exit(1 if bad else 0)
