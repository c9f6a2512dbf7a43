"""Goal-list soak WITHOUT a GPU: random goal lists on the PR2-like robot -- every link goal type on any link of the chains, several goals per link, gene-only
goals primary and secondary, JointVariableGoals (which reorder the active variables), fixed joints, the three groups -- in the host simulator (tests/hostsim)
against the CPU oracle: fitness (exact and linearised), approximator tables, success test and whole solves bit for bit.  The link goals are listed in the
order the chain walk completes their links and the gene-only goals behind them: the order in which device and reference add the same sum (DESIGN.md
section 4).  usage: python tools/goal_fuzz_hostsim.py [cases] [seed]   (seconds per case; exit code 1 on a mismatch)"""
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
                        MinDistanceGoal, MinimalDisplacementGoal, OrientationGoal, PlaneGoal, PoseGoal, PositionGoal, ProblemTemplate, RegularizationGoal, SideGoal,
                        abi, pr2_like, solver)
from oracle import orc  # noqa: E402


def unit(rng, n):
    v = rng.normal(size=n)
    return tuple(v / np.linalg.norm(v))


def link_goal(rng, link):
    w = float(rng.choice([0.2, 0.5, 1.0, 1.7]))
    p = tuple(rng.normal(size=3) * 0.4)
    k = int(rng.integers(11))
    if k == 0:
        return PositionGoal(link, p, weight=w)
    if k == 1:
        return OrientationGoal(link, unit(rng, 4), weight=w)
    if k == 2:
        return PoseGoal(link, p, unit(rng, 4), weight=w)
    if k == 3:
        return LookAtGoal(link, unit(rng, 3), p, weight=w)
    if k == 4:
        return MaxDistanceGoal(link, p, float(rng.uniform(0.1, 0.6)), weight=w)
    if k == 5:
        return MinDistanceGoal(link, p, float(rng.uniform(0.1, 0.6)), weight=w)
    if k == 6:
        return LineGoal(link, p, unit(rng, 3), weight=w)
    if k == 7:
        return PlaneGoal(link, p, unit(rng, 3), weight=w)
    if k == 8:
        return SideGoal(link, unit(rng, 3), unit(rng, 3), weight=w)
    if k == 9:
        return DirectionGoal(link, unit(rng, 3), unit(rng, 3), weight=w)
    return ConeGoal(link, unit(rng, 3), unit(rng, 3), float(rng.uniform(0.1, 0.8)), weight=w, position=p if rng.random() < 0.5 else None, position_weight=0.5)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    lib = solver.load_library(os.path.join(ROOT, "tests", "hostsim", "libbioik_hostsim.so"))
    orc.set_trig_mode(1)
    model = pr2_like()
    arm = {"right_arm": [n for n in model.link_names if n.startswith("r_")], "left_arm": [n for n in model.link_names if n.startswith("l_")]}
    arm["all"] = arm["right_arm"] + arm["left_arm"]
    joints = {g: [n.replace("_link", "_joint") for n in v if n.replace("_link", "_joint") not in ("r_upper_arm_joint", "l_upper_arm_joint", "r_forearm_joint", "l_forearm_joint")] for g, v in arm.items()}
    bad = 0
    for case in range(n_cases):
        group = str(rng.choice(["right_arm", "left_arm", "all"]))
        links = sorted(rng.choice(len(arm[group]), size=int(rng.integers(1, 4)), replace=False))
        goals = []
        for li in links:  # (ascending link index = the order the walk completes them)
            for _ in range(int(rng.choice([1, 1, 2, 3]))):
                goals.append(link_goal(rng, arm[group][li]))
        for _ in range(int(rng.integers(0, 4))):  # gene-only goals, behind the link goals
            k = int(rng.integers(5))
            w = float(rng.choice([0.1, 0.4, 0.9]))
            sec = bool(rng.random() < 0.5)
            if k == 0:
                goals.append(JointVariableGoal(str(rng.choice(joints[group])), float(rng.normal() * 0.5), weight=w, secondary=sec))
            elif k == 1:
                g = RegularizationGoal(weight=w)
                g.secondary_ = sec
                goals.append(g)
            else:
                goals.append((MinimalDisplacementGoal, AvoidJointLimitsGoal, CenterJointsGoal)[k - 2](weight=w, secondary=sec))
        fixed = [str(rng.choice(joints[group]))] if rng.random() < 0.25 else []
        desc = "%s fixed=%s | %s" % (group, fixed, " ".join("%s%s" % (type(g).__name__.replace("Goal", ""), "*" if g.secondary_ else "") for g in goals))
        try:
            t = ProblemTemplate(model, group, goals, fixed_joints=fixed)
            h, o = solver.HipSolver(t, lib=lib), orc.Oracle(t)
            if o.D == 0:  # (the fixed joint was the only one that moves a goal's link)
                print("%-3d skip %s (no active variable)" % (case, desc), flush=True)
                continue
            pc.function_level(h, o, model, np.random.default_rng(case), n=24, exact_bits=True)
            mode = str(rng.choice(["bio2", "bio2_memetic", "bio2_memetic_l"]))
            fk = int(rng.choice([abi.FK_EXACT, abi.FK_LINEAR]))
            pc.trajectory(h, o, t, n=2, pop=int(rng.choice([8, 16, 33])), steps_list=(int(rng.choice([1, 2, 3])),), mode=mode, fk_mode=fk, seed=case)
            print("%-3d ok   %s" % (case, desc), flush=True)
        except solver.BioIKError as e:
            print("%-3d skip %s (%s)" % (case, desc, e), flush=True)
        except AssertionError as e:
            import traceback
            bad += 1
            print("%-3d BAD  %s: %s @ %s" % (case, desc, e, traceback.format_exc().splitlines()[-3].strip()), flush=True)
    print("%d cases, %d mismatches" % (n_cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
