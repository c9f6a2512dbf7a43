"""The launcher's measured mapping choice (bioik_hip.hip: solve_dispatch) on the BASELINE configurations and on a problem outside every fitted threshold
(a 12-joint chain, 64 children per species): the table every handle's first chip-filling call produces (BIOIK_SOLVE_REPORT), and the time of a call under
the rules alone against the time under the choice."""
import os, sys, time
os.environ["BIOIK_SOLVE_REPORT"] = "1"
import numpy as np
sys.path.insert(0, '.')
from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
cases = [("c2", ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")]), 128, 64, 4096),
         ("c3", ProblemTemplate(pr2_like(), "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), MinimalDisplacementGoal()]), 128, 64, 4096),
         ("c4", ProblemTemplate(snake(31), "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()]), 512, 32, 4096),
         ("snake12_pop64", ProblemTemplate(snake(12), "snake", [PoseGoal("tip")]), 64, 48, 4096),
         ("snake12_pop64_linear", ProblemTemplate(snake(12), "snake", [PoseGoal("tip")]), 64, 48, 4096)]
for name, t, pop, steps, n in cases:
    fk = abi.FK_LINEAR if name.endswith("linear") else abi.FK_EXACT
    p = abi.default_solve_params(population=pop, max_steps=steps, random_seed=1, fk_mode=fk)
    out = {}
    for tune in ("0", "1"):
        os.environ["BIOIK_SOLVE_AUTOTUNE"] = tune
        h = HipSolver(t, device=0)
        seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, n, seed=5)
        print("== %s, BIOIK_SOLVE_AUTOTUNE=%s" % (name, tune), file=sys.stderr, flush=True)
        os.environ["BIOIK_SOLVE_REPORT"] = "1"
        r = h.solve_batch(p, seeds, params)  # (the first chip-filling call of the handle: times the presets when the switch is on)
        os.environ.pop("BIOIK_SOLVE_REPORT")
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); r2 = h.solve_batch(p, seeds, params); ts.append(time.perf_counter() - t0)
        assert all(np.array_equal(a, b) for a, b in zip(r, r2))
        out[tune] = (r, 1e3 * float(np.median(ts)))
    same = all(np.array_equal(a, b) for a, b in zip(out["0"][0], out["1"][0]))
    print("%s: host-pointer call under the rules %.3f ms, under the measured choice %.3f ms, results identical: %s, success %.4f" % (name, out["0"][1], out["1"][1], same, out["1"][0][2].mean()), flush=True)
