"""Randomised parity soak on a GPU box: many (fixture, population, mode, phenotype model, islands, mapping) combinations, whole solves
compared bit for bit with the CPU oracle run on the same random streams.  Not part of the test suite (minutes of oracle time);
prints one line per case and a summary, exit code 1 on any mismatch.  usage: python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bio_ik_amd import (AvoidJointLimitsGoal, CenterJointsGoal, JointVariableGoal, MinimalDisplacementGoal, PoseGoal, PositionGoal, ProblemTemplate, abi, pr2_like,  # noqa: E402
                        snake)
from bio_ik_amd.solver import HipSolver  # noqa: E402
from bio_ik_amd.workload import make_queries  # noqa: E402
from conftest import mimic_robot, mobile_robot  # noqa: E402
from oracle import orc  # noqa: E402


def templates():
    pr2 = pr2_like()
    out = {
        "c2": ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")]),
        "c2+pos": ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link"), PositionGoal("r_elbow_flex_link", weight=0.1)]),
        "c3": ProblemTemplate(pr2, "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), MinimalDisplacementGoal()]),
        # (round 5: several secondary goals at once; the JointVariableGoal puts its variable in front of the chain's, so the genes do not follow the ops)
        "c2+sec": ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link"), MinimalDisplacementGoal(weight=0.7), AvoidJointLimitsGoal(weight=0.3),
                                                     JointVariableGoal("r_elbow_flex_joint", -1.0, weight=0.5, secondary=True), CenterJointsGoal(weight=0.2)]),
        "c2+sec2": ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link"), MinimalDisplacementGoal(weight=0.7), AvoidJointLimitsGoal(weight=0.3), CenterJointsGoal(weight=0.2)]),
        "c4": ProblemTemplate(snake(31), "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()]),
        "snake12": ProblemTemplate(snake(12), "snake", [PoseGoal("tip")]),
        "mimic": ProblemTemplate(mimic_robot(), "arm", [PoseGoal("tool"), MinimalDisplacementGoal(weight=0.5)]),
        "floating": ProblemTemplate(mobile_robot("floating"), "whole", [PoseGoal("tool"), PositionGoal("base", weight=0.2)]),
        "planar": ProblemTemplate(mobile_robot("planar"), "whole", [PoseGoal("tool"), PositionGoal("base", weight=0.2)]),
    }
    return out


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    orc.set_trig_mode(1)
    T = templates()
    H = {k: HipSolver(t, device=0) for k, t in T.items()}
    O = {k: orc.Oracle(t) for k, t in T.items()}
    bad = 0
    t_start = time.time()
    for case in range(n_cases):
        name = rng.choice(list(T))
        t, h, o = T[name], H[name], O[name]
        pop = int(rng.choice([8, 16, 24, 33, 64, 70, 128, 200, 512]))
        mode = str(rng.choice(["bio2", "bio2_memetic", "bio2_memetic_l"]))
        fk = int(rng.choice([abi.FK_EXACT, abi.FK_LINEAR]))
        if name in ("floating", "planar") and mode != "bio2":
            mode = "bio2"  # their Jacobian columns go through acos / sqrt of two different math libraries: bit-exact only without them
            fk = abi.FK_EXACT
        islands = int(rng.choice([1, 1, 2, 3]))
        steps = int(rng.choice([1, 2, 5, 9]))
        if name not in ("floating", "planar") and rng.random() < 0.12:  # the gradient family (round 3: gd_r, islands = the _N solver names)
            mode, fk = str(rng.choice(["gd", "gd_r", "gd_c"])), abi.FK_EXACT
            islands, steps = int(rng.choice([1, 2, 4, 8])), int(rng.choice([1, 5, 20, 40]))
        n = int(rng.choice([3, 17, 40]))
        env = {}
        r = rng.random()
        if r < 0.15:
            env = {"BIOIK_SOLVE_THREADS": "64"}
        elif r < 0.3:
            env = {"BIOIK_SOLVE_THREADS": "128"}
        elif r < 0.4:
            env = {"BIOIK_SOLVE_THREADS": "256"}
        elif r < 0.5:
            env = {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1"}
        elif r < 0.6:
            env = {"BIOIK_SOLVE_GENERAL": "1"}
        elif r < 0.7:  # both species on one wavefront, children computed in pairs: with a secondary goal and exact FK the joint walk (k_solve_lean_clj)
            env = {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_COLUMNLESS": "2"}
            if rng.random() < 0.3:
                env["BIOIK_SOLVE_NO_JOINT"] = "1"
        if rng.random() < 0.2:
            env["BIOIK_SOLVE_STORE_CHILDREN"] = "0"
        elif rng.random() < 0.35 and "BIOIK_SOLVE_GENERAL" not in env and "BIOIK_SOLVE_COLUMNLESS" not in env:  # children computed where they are read (round 2), singly or in pairs
            env["BIOIK_SOLVE_COLUMNLESS"] = str(rng.choice(["1", "2"]))
        if rng.random() < 0.3:  # the solve split over two or more launches (round 2), under whatever mapping was drawn above
            env["BIOIK_SOLVE_TWO_PHASE"] = str(rng.choice(["1", "2", "3", "1,2", "2,4,6", "init"]))
        elif rng.random() < 0.2:  # every unit leaves its first launch at its own step (round 4: the hand-over when the chip runs empty, as its test pattern)
            env["BIOIK_SOLVE_DRAIN_TEST"] = str(rng.choice(["2", "5", "9"]))
        if rng.random() < 0.4:  # round 5: small launches of PoseGoal-class problems run k_solve_lean_cl4's helped build by default: half of the draws keep the plain one
            env["BIOIK_SOLVE_HELPED"] = "0"
        if rng.random() < 0.15:  # the pre-selection's sort keys give up so many bits that its exact path runs in most generations (round 4)
            env["BIOIK_SOLVE_SORT_KEY_DROP"] = str(rng.choice(["30", "44", "51"]))
        if rng.random() < 0.3:  # round 5: the kernels of the 128-register budget SELECT the pre-selection's survivors; a third of the draws keep the sort of all children
            env["BIOIK_SOLVE_PRESELECT"] = "0"
        kw = {"no_wipeout": int(rng.random() < 0.2), "schedule": int(rng.random() < 0.25)}  # (schedule: BIOIK_SCHEDULE_THROUGHPUT where its mapping exists)
        if islands > 1 and rng.random() < 0.5:
            kw["island_sync"] = 1  # (round 4: the islands of a query stop once one of them has passed)
        seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=int(rng.integers(1 << 30)), kind=str(rng.choice(["global", "tracking"])))
        p = abi.default_solve_params(population=pop, max_steps=steps, random_seed=int(rng.integers(1 << 30)), mode=mode, fk_mode=fk, islands=islands, **kw)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            sb = h.solve_batch(p, seeds, params)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        sa = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=8)
        ok = np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1]) and np.array_equal(sa[2], sb[2]) and np.array_equal(sa[3], sb[3])
        bad += 0 if ok else 1
        print("%-3d %-9s pop=%-3d %-15s fk=%d islands=%d steps=%d n=%-2d sched=%d %-60s %s" %
              (case, name, pop, mode, fk, islands + 10 * kw.get("island_sync", 0), steps, n, kw["schedule"], str(env), "ok" if ok else "MISMATCH max|dx|=%g" % np.abs(sa[0] - sb[0]).max()), flush=True)
    print("%d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t_start))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
