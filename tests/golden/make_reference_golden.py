"""Generate tests/golden/reference_*.npz from the REFERENCE'S OWN code (oracle/_ref/libbioik_ref.so = the reference's
include/bio_ik/*.h, src/forward_kinematics.h, src/problem.cpp, src/ik_base.h, src/ik_evolution_2.cpp compiled
unmodified against oracle/ref_shim).  Run where /root/reference exists:   python tests/golden/make_reference_golden.py
The fixtures let every later checkout (and the GPU box) pin the oracle against reference outputs without the reference tree."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bio_ik_amd import AvoidJointLimitsGoal, BalanceGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake  # noqa: E402
from bio_ik_amd.workload import make_queries  # noqa: E402
from conftest import balance_robot, gnarly_goals, gnarly_robot, random_configuration  # noqa: E402
from oracle import orc, ref  # noqa: E402


def templates():
    pr2, sn, gn = pr2_like(), snake(31), gnarly_robot()
    return {
        "c2": ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")]),
        "c3": ProblemTemplate(pr2, "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), MinimalDisplacementGoal()]),
        "c4": ProblemTemplate(sn, "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()]),
        "gnarly": ProblemTemplate(gn, "body", gnarly_goals()),
        "balance": ProblemTemplate(balance_robot(), "body", [PoseGoal("a_tool"), BalanceGoal((0.02, -0.01, 0.0), weight=0.8)]),
    }


def main():
    rng = np.random.default_rng(2026)
    out = {}
    for name, t in templates().items():
        r = ref.Reference(t)
        o = orc.Oracle(t)  # only for the query generator's FK (inputs), never for expected outputs
        n = 24
        vars_ = random_configuration(t.model, rng, n)
        seed = vars_[0]
        genes = vars_[:, r.active_variables]
        raw = rng.normal(size=r.P)
        par = r.canonical_params(raw)
        base = genes[1]
        near = base + 0.01 * rng.normal(size=(n, r.D))
        pe, se = r.fitness(abi.FK_EXACT, seed, raw, genes)
        pl, sl = r.fitness(abi.FK_LINEAR, seed, raw, near, base)
        bt, lin = r.approx_eval(seed, base, near)
        out.update({name + "/vars": vars_, name + "/fk": r.fk(vars_), name + "/robot_info": r.robot_info(), name + "/params_raw": raw,
                    name + "/params": par, name + "/primary_exact": pe, name + "/secondary_exact": se, name + "/near": near,
                    name + "/primary_linear": pl, name + "/base_tips": bt, name + "/linear_frames": lin,
                    name + "/check": r.check(seed, raw, genes), name + "/active_variables": r.active_variables, name + "/tip_links": r.tip_links})
        if name in ("gnarly", "balance"):
            continue
        # whole trajectories of the reference solver (reference RNG: std::minstd_rand + tables, seed 5), 3 queries x 10 steps
        seeds, params, _ = make_queries(t, r.active_variables, o.fk_genes, 3, seed=3)
        for mode in ("bio2_memetic", "bio2", "bio2_memetic_l"):
            p = abi.default_solve_params(population=16, fk_mode=abi.FK_LINEAR, mode=mode, random_seed=5, max_steps=1000)
            rr = ref.Reference(t, p)
            sols, fits, sucs, cans = [], [], [], []
            for q in range(3):
                s, f, k = rr.solve_steps(seeds[q], params[q], 10)
                sols.append(s), fits.append(f), sucs.append(k), cans.append(rr.canonical_params(params[q]))
            out.update({"%s/%s/solutions" % (name, mode): np.array(sols), "%s/%s/fitness" % (name, mode): np.array(fits),
                        "%s/%s/success" % (name, mode): np.array(sucs), "%s/%s/params" % (name, mode): np.array(cans)})
        out[name + "/traj_seeds"] = seeds
    # L1 frame algebra
    fr = rng.normal(size=(64, 3, 7))
    fr[:, :, 3:] /= np.linalg.norm(fr[:, :, 3:], axis=2, keepdims=True)
    out["frames/in"] = fr
    out["frames/concat"] = np.array([ref.frame_concat(a, b) for a, b, _ in fr])
    out["frames/invert"] = np.array([ref.frame_invert(a) for a, _, _ in fr])
    out["frames/change"] = np.array([ref.frame_change(a, b, c) for a, b, c in fr])
    out["frames/twist"] = np.array([ref.frame_twist(a, b) for a, b, _ in fr])
    out["frames/quat_mul_vec"] = np.array([ref.quat_mul_vec(a[3:], b[:3]) for a, b, _ in fr])
    np.savez_compressed(os.path.join(HERE, "reference_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_golden.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
