"""Serial-chain soak WITHOUT a GPU: random chains of 2 - 40 revolute / continuous joints (any axis, any offset, unrotated origins: the joint program folds
nothing there, so the default program is the reference's arithmetic), populations around the kernels' lane counts (32 ... 512), one to three steps, both
schedules, one or two islands, with and without a secondary goal -- the problems the launcher's rules hand to the kernels compiled for ONE lane mapping
(k_solve_lean_cl4 / cl4h / clj4 / cl64w4 / lin) and to the generic ones around their thresholds.  Host simulator against the CPU oracle, whole solves bit for
bit; prints which kernel ran.  usage: python tools/chain_fuzz_hostsim.py [cases] [seed]   (exit code 1 on a mismatch)"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hostsim"), "-s"], check=True)
os.environ["BIOIK_SOLVE_REPORT"] = "1"
from bio_ik_amd import (AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, PositionGoal, ProblemTemplate, RobotModel, abi, solver)  # noqa: E402
from bio_ik_amd.workload import make_queries  # noqa: E402
from oracle import orc  # noqa: E402


def chain(rng, n, case):
    m = RobotModel("c%d" % case)
    m.add_link("base")
    prev = "base"
    for i in range(n):
        axis = rng.normal(size=3) if rng.random() < 0.4 else np.eye(3)[int(rng.integers(3))]
        kw = {"lower": float(-rng.uniform(0.3, 2.5)), "upper": float(rng.uniform(0.3, 2.5))} if rng.random() < 0.85 else {}
        m.add_link("s%d" % i, prev, "j%d" % i, "revolute" if kw else "continuous", xyz=tuple(rng.normal(size=3) * (1.0 / n)), axis=tuple(axis / np.linalg.norm(axis)),
                   velocity=float(rng.uniform(0.5, 3.0)), **kw)
        prev = "s%d" % i
    m.add_link("tip", prev, "tip_joint", "fixed", xyz=(0.05, 0.0, 0.02))
    m.add_group("g", chain=("base", "tip"))
    return m


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    lib = solver.load_library(os.path.join(ROOT, "tests", "hostsim", "libbioik_hostsim.so"))
    orc.set_trig_mode(1)
    bad, kernels = 0, {}
    for case in range(n_cases):
        n = int(rng.choice([2, 3, 5, 7, 7, 9, 12, 16, 24, 31, 40]))
        model = chain(rng, n, case)
        goals = [PoseGoal("tip") if rng.random() < 0.7 else PositionGoal("tip")]
        sec = int(rng.integers(3))
        if sec == 1:
            goals.append(MinimalDisplacementGoal(weight=float(rng.choice([0.3, 1.0]))))
        elif sec == 2:
            goals.append(AvoidJointLimitsGoal(weight=float(rng.choice([0.3, 1.0]))))
        pop = int(rng.choice([32, 64, 100, 128, 129, 200, 256, 512]))
        mode = str(rng.choice(["bio2", "bio2_memetic", "bio2_memetic", "bio2_memetic_l"]))
        fk = abi.FK_EXACT if rng.random() < 0.8 else abi.FK_LINEAR
        nq, steps = int(rng.choice([1, 2, 5])), int(rng.choice([1, 2, 3]))
        islands = int(rng.choice([1, 1, 2]))
        sched = abi.SCHEDULE_THROUGHPUT if rng.random() < 0.3 else abi.SCHEDULE_LATENCY
        # the launcher's own choice for a small call, or one of the mappings it gives large ones (its rules look at the number of units; a host simulator cannot fill a chip)
        env = [{}, {}, {"BIOIK_SOLVE_HELPED": "0"}, {"BIOIK_SOLVE_HELPED": "0", "BIOIK_SOLVE_FOUR_WAVES": "1"},
               {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_COLUMNLESS": "2"},
               {"BIOIK_SOLVE_THREADS": "64", "BIOIK_SOLVE_SPECIES_PARALLEL": "1", "BIOIK_SOLVE_COLUMNLESS": "2", "BIOIK_SOLVE_THREE_WAVES": "1"},
               {"BIOIK_SOLVE_THREADS": "128", "BIOIK_SOLVE_COLUMNLESS": "1"}, {"BIOIK_SOLVE_THREADS": "256"}, {"BIOIK_SOLVE_TWO_PHASE": "1"}, {"BIOIK_SOLVE_DRAIN_TEST": "2"}][int(rng.integers(10))]
        desc = "%2d joints pop %3d %-14s fk %d n %d steps %d islands %d sched %d %s %s" % (n, pop, mode, fk, nq, steps, islands, sched, " ".join(type(g).__name__.replace("Goal", "") for g in goals),
                                                                                          " ".join("%s=%s" % (k.replace("BIOIK_SOLVE_", ""), v) for k, v in env.items()))
        t = ProblemTemplate(model, "g", goals)
        try:
            h, o = solver.HipSolver(t, lib=lib), orc.Oracle(t)
            seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, nq, seed=case)
            p = abi.default_solve_params(population=pop, max_steps=steps, random_seed=7, mode=mode, fk_mode=fk, islands=islands, schedule=sched)
            with tempfile.TemporaryFile(mode="w+") as err:  # (the launcher's report goes to stderr: which kernel ran)
                fd = os.dup(2)
                os.dup2(err.fileno(), 2)
                os.environ.update(env)
                try:
                    got = h.solve_batch(p, seeds, params)
                finally:
                    for k in env:
                        os.environ.pop(k)
                    os.dup2(fd, 2)
                    os.close(fd)
                err.seek(0)
                ran = sorted(set(re.findall(r"launch: (k_\w+)", err.read())))
            want = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=4)
            ok = all(np.array_equal(a, b) for a, b in zip(want, got))
            for k in ran:
                kernels[k] = kernels.get(k, 0) + 1
            bad += 0 if ok else 1
            print("%-3d %s %s  [%s]" % (case, "ok " if ok else "BAD", desc, ", ".join(ran)), flush=True)
        except solver.BioIKError as e:
            print("%-3d skip %s (%s)" % (case, desc, e), flush=True)
    print("kernels:", kernels)
    print("%d cases, %d mismatches" % (n_cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
