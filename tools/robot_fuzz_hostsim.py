"""Robot soak WITHOUT a GPU: random kinematic TREES -- revolute, continuous, prismatic and fixed joints anywhere, rotated origins, oblique axes, mimic joints,
branches at any depth, tips on inner links, on fixed links and off the root, fixed_joints -- with goals listed in walk order, in the host simulator
(tests/hostsim) against the CPU oracle.  Twice per robot: the unfolded joint program (BIOIK_COMPILE_EXACT=1, bioik_compile.cpp), where FK, fitness, tables,
success test and a whole solve must be the oracle's bit for bit on ANY robot, and the default (folded) program, which must agree to rounding (1e-12).
ROBOT_FUZZ_GRADIENT=1: every third robot is solved by a point solver of the gradient family (gd / gd_r / gd_c) instead; ROBOT_FUZZ_BIG=1: 12 - 30 links, up to six tips; ROBOT_FUZZ_PLAIN=1: trees whose default program folds exactly (unrotated origins, no prismatic joint, fixed links without offset) -- the DEFAULT program bit for bit, populations 16 ... 200; ROBOT_FUZZ_BALANCE=1: links with mass and a BalanceGoal (whose sum over the links the device takes in walk order: agreement to rounding, DESIGN.md section 7 -- the strict comparison of this tool then reports it).  (Floating / planar joints are not drawn:
their unbounded variables need a sampler of their own; tests/test_*_parity.py: test_floating_and_planar_joints_anywhere covers them on fixtures.)
usage: python tools/robot_fuzz_hostsim.py [cases] [seed]   (seconds per case; exit code 1 on a mismatch)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hostsim"), "-s"], check=True)
import parity_cases as pc  # noqa: E402
from bio_ik_amd import (AvoidJointLimitsGoal, CenterJointsGoal, ConeGoal, DirectionGoal, JointVariableGoal, LineGoal, LookAtGoal, MaxDistanceGoal,  # noqa: E402
                        MinDistanceGoal, MinimalDisplacementGoal, OrientationGoal, PlaneGoal, PoseGoal, PositionGoal, ProblemTemplate, RegularizationGoal, RobotModel,
                        SideGoal, abi, solver)
from oracle import orc  # noqa: E402


def unit(rng, n):
    v = rng.normal(size=n)
    return tuple(v / np.linalg.norm(v))


def random_robot(rng, case):
    m = RobotModel("r%d" % case)
    m.add_link("l0")
    n = int(rng.integers(4, 15)) if not os.environ.get("ROBOT_FUZZ_BIG") else int(rng.integers(12, 31))  # (ROBOT_FUZZ_BIG=1: 12 - 30 links, up to six tips; ROBOT_FUZZ_PLAIN=1: trees whose default program folds exactly (unrotated origins, no prismatic joint, fixed links without offset) -- the DEFAULT program bit for bit, populations 16 ... 200; ROBOT_FUZZ_BALANCE=1: links with mass and a BalanceGoal (whose sum over the links the device takes in walk order: agreement to rounding, DESIGN.md section 7 -- the strict comparison of this tool then reports it).
    joints, moving, mimicable = [], [], []
    for i in range(1, n):
        parent = "l%d" % (i - 1 if rng.random() < 0.7 else int(rng.integers(0, i)))
        kind = str(rng.choice(["revolute", "revolute", "revolute", "continuous", "prismatic", "fixed"]))
        xyz = tuple(rng.normal(size=3) * 0.15) if rng.random() < 0.8 else (0.0, 0.0, 0.0)
        rpy = tuple(rng.normal(size=3) * 0.6) if rng.random() < 0.5 else (0.0, 0.0, 0.0)
        if os.environ.get("ROBOT_FUZZ_PLAIN"):  # (ROBOT_FUZZ_PLAIN=1: trees whose DEFAULT joint program folds exactly -- no rotated origin, no prismatic joint, fixed links without offset)
            rpy = (0.0, 0.0, 0.0)
            if kind == "prismatic":
                kind = "revolute"
            if kind == "fixed":
                xyz = (0.0, 0.0, 0.0)
        axis = unit(rng, 3) if rng.random() < 0.5 else tuple(np.eye(3)[int(rng.integers(3))])
        kw = {}
        if kind in ("revolute", "prismatic"):
            lo, hi = sorted(rng.normal(size=2) * (1.5 if kind == "revolute" else 0.2))
            kw = {"lower": float(lo - 0.1), "upper": float(hi + 0.1)}
        if kind != "fixed":
            kw["velocity"] = float(rng.uniform(0.3, 3.0))
            if moving and kind in ("revolute", "prismatic") and rng.random() < 0.12:
                kw["mimic"] = (str(rng.choice(mimicable)), float(rng.choice([1.0, -0.5, 2.0])), float(rng.choice([0.0, 0.1])))  # (also of a joint that mimics another)
        if os.environ.get("ROBOT_FUZZ_BALANCE") and rng.random() < 0.6:  # (ROBOT_FUZZ_BALANCE=1: links with mass, a BalanceGoal among the goals)
            kw["mass"], kw["com"] = float(rng.uniform(0.2, 3.0)), tuple(rng.normal(size=3) * 0.05)
        m.add_link("l%d" % i, parent, "j%d" % i, kind, xyz=xyz, rpy=rpy, axis=axis, **kw)
        if kind != "fixed":
            joints.append("j%d" % i)
            if kind in ("revolute", "prismatic"):
                mimicable.append("j%d" % i)
                if "mimic" not in kw:
                    moving.append("j%d" % i)
    return m, joints, n


def walk_order(model, tips):
    """the tips in the order the chain walk completes them (bioik_compile.cpp: the links are scheduled chain by chain in the order of the tips, a tip is complete with
    the op of its nearest moving ancestor -- a tip behind fixed links only hangs off the root and is complete before the walk starts)"""
    tips = list(tips)
    for _ in range(len(tips) + 1):
        schedule = []
        for t in tips:
            chain, l = [], t
            while l >= 0:
                chain.append(l)
                l = model.link_parent[l]
            for l in reversed(chain):
                if l not in schedule:
                    schedule.append(l)
        ops = [l for l in schedule if model.joint_type[l] != 0]

        def src(t):
            l = t
            while l >= 0 and model.joint_type[l] == 0:
                l = model.link_parent[l]
            return ops.index(l) if l >= 0 else -1
        again = sorted(tips, key=lambda t: (src(t), tips.index(t)))
        if again == tips:
            break
        tips = again
    return tips


def whole_solve(h, o, t, pop, steps, mode, fk, case):
    """parity_cases.trajectory with NaNs compared as equal: a goal that is met exactly makes the quadratic line search divide 0 by 0 (ik_evolution_2.cpp:498-539),
    in the reference as here, and the NaN genes that follow must then be the same ones on both sides"""
    from bio_ik_amd.workload import make_queries
    seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 2, seed=case)
    p = abi.default_solve_params(population=pop, max_steps=steps, random_seed=11, mode=mode, fk_mode=fk, islands=1 + case % 2)
    # (Until round 5 a solve in which an infinite step of the line search had put a joint WITHOUT limits at +-DBL_MAX was set aside here when the two sides parted:
    # 2 of 8000.  Round 6: such a candidate is no candidate on either side (quirk Q7, BIOIK_CANDIDATE_BOUND), and the two expressions that parted behind it are
    # known -- acos of a NaN (now tf2Acos's on both sides) and the struck-out zero terms of axis-aligned joints at sin / cos = inf.  Every solve is compared.)
    sa = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=4)
    sb = h.solve_batch(p, seeds, params)
    for a, b in zip(sa, sb):
        assert np.array_equal(a, b, equal_nan=True), "whole solves differ: %g" % np.nanmax(np.abs(np.asarray(a, float) - np.asarray(b, float)))
    return "  [NaN genes, the same on both sides]" if np.isnan(sa[0]).any() else ""


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    lib = solver.load_library(os.path.join(ROOT, "tests", "hostsim", "libbioik_hostsim.so"))
    orc.set_trig_mode(1)
    bad = skipped = 0
    for case in range(n_cases):
        model, joints, n = random_robot(rng, case)
        if not joints:
            continue
        tips = walk_order(model, sorted(int(t) for t in rng.choice(np.arange(1, n), size=min(int(rng.integers(1, 7 if os.environ.get("ROBOT_FUZZ_BIG") else 4)), n - 1), replace=False)))
        model.add_group("g", joints=joints, tips=["l%d" % t for t in tips])
        goals = []
        for t in tips:  # (the order in which the walk completes them: the order in which device and reference add the same sum)
            for _ in range(int(rng.choice([1, 1, 2]))):
                link, w, p = "l%d" % t, float(rng.choice([0.3, 1.0, 1.6])), tuple(rng.normal(size=3) * 0.3)
                k = int(rng.integers(11))  # (the last four call acos: the host simulator shares the oracle's)
                goals.append([PositionGoal(link, p, weight=w), OrientationGoal(link, unit(rng, 4), weight=w), PoseGoal(link, p, unit(rng, 4), weight=w),
                              MaxDistanceGoal(link, p, 0.3, weight=w), MinDistanceGoal(link, p, 0.3, weight=w), LineGoal(link, p, unit(rng, 3), weight=w),
                              PlaneGoal(link, p, unit(rng, 3), weight=w), LookAtGoal(link, unit(rng, 3), p, weight=w), SideGoal(link, unit(rng, 3), unit(rng, 3), weight=w),
                              DirectionGoal(link, unit(rng, 3), unit(rng, 3), weight=w), ConeGoal(link, unit(rng, 3), unit(rng, 3), 0.4, weight=w)][k])
        if os.environ.get("ROBOT_FUZZ_BALANCE") and sum(model.link_mass) > 0:
            from bio_ik_amd import BalanceGoal
            goals.append(BalanceGoal(tuple(rng.normal(size=3) * 0.1), weight=float(rng.choice([0.4, 1.0]))))
        for _ in range(int(rng.integers(0, 3))):
            k, w, sec = int(rng.integers(5)), float(rng.choice([0.1, 0.5])), bool(rng.random() < 0.5)
            if k == 0:
                goals.append(JointVariableGoal(str(rng.choice(joints)), float(rng.normal() * 0.3), weight=w, secondary=sec))
            elif k == 1:
                g = RegularizationGoal(weight=w)
                g.secondary_ = sec
                goals.append(g)
            else:
                goals.append((MinimalDisplacementGoal, AvoidJointLimitsGoal, CenterJointsGoal)[k - 2](weight=w, secondary=sec))
        fixed = [str(rng.choice(joints))] if rng.random() < 0.2 else []
        desc = "%d links, tips %s, fixed %s | %s" % (n, list(tips), fixed, " ".join("%s%s" % (type(g).__name__.replace("Goal", ""), "*" if g.secondary_ else "") for g in goals))
        mode = str(rng.choice(["bio2", "bio2_memetic", "bio2_memetic_l"]))
        fk = int(rng.choice([abi.FK_EXACT, abi.FK_LINEAR]))
        pop, steps = int(rng.choice([8, 16, 33])), int(rng.choice([1, 2, 3]))
        if os.environ.get("ROBOT_FUZZ_GRADIENT") and case % 3 == 0:  # (the gradient family's point solvers on the same trees)
            mode, fk, steps = str(("gd", "gd_r", "gd_c")[(case // 3) % 3]), abi.FK_EXACT, int(rng.choice([1, 5, 20]))
        try:
            t = ProblemTemplate(model, "g", goals, fixed_joints=fixed)
            try:
                o = orc.Oracle(t)
            except orc.OracleError as e:  # (what the reference refuses -- a goal on the variable of a mimic joint -- the device must refuse too)
                try:
                    solver.HipSolver(t, lib=lib)
                    bad += 1
                    print("%-3d BAD  %s: the oracle refuses (%s), the device does not" % (case, desc, e), flush=True)
                except solver.BioIKError as e2:
                    skipped += 1
                    print("%-3d skip %s (both refuse: %s | %s)" % (case, desc, e, e2), flush=True)
                continue
            if o.D == 0:
                skipped += 1
                print("%-3d skip %s (no active variable)" % (case, desc), flush=True)
                continue
            if os.environ.get("ROBOT_FUZZ_PLAIN"):  # the default program itself, bit for bit, populations up to the lane counts of the kernels compiled for one mapping
                os.environ["BIOIK_COMPILE_EXACT"] = "0"
                h = solver.HipSolver(t, lib=lib)
                pc.function_level(h, o, model, np.random.default_rng(case), n=16, exact_bits=True)
                nan_note = whole_solve(h, o, t, int(rng.choice([16, 64, 128, 200])), steps, mode, fk, case)
                print("%-3d ok   %s%s" % (case, desc, nan_note), flush=True)
                continue
            os.environ["BIOIK_COMPILE_EXACT"] = "1"
            h = solver.HipSolver(t, lib=lib)
            pc.function_level(h, o, model, np.random.default_rng(case), n=16, exact_bits=True)
            nan_note = whole_solve(h, o, t, pop, steps, mode, fk, case)
            os.environ["BIOIK_COMPILE_EXACT"] = "0"
            h2 = solver.HipSolver(t, lib=lib)
            pc.function_level(h2, o, model, np.random.default_rng(case), n=16, frame_tol=1e-12, fit_rtol=1e-9)
            print("%-3d ok   %s%s" % (case, desc, nan_note), flush=True)
        except solver.BioIKError as e:
            skipped += 1
            print("%-3d skip %s (%s)" % (case, desc, e), flush=True)
        except AssertionError as e:
            import traceback
            bad += 1
            print("%-3d BAD  [EXACT=%s] %s: %s @ %s" % (case, os.environ["BIOIK_COMPILE_EXACT"], desc, e, traceback.format_exc().splitlines()[-3].strip()), flush=True)
    print("%d cases, %d skipped (unsupported by the device, or no active variable), %d mismatches" % (n_cases, skipped, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
