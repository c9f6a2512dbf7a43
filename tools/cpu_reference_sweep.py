"""The reference's OWN bio2_memetic (oracle/_ref: its sources compiled unmodified with its Release flags, its population of
2 x (2 + 16), its linearised phenotypes, one island = one thread) on the BASELINE.json configurations C2 / C3 / C4: the CPU
figures next to tools/config_sweep.py.  Budget form of the island loop: success test after every step, at most `budget` steps.
usage: python tools/cpu_reference_sweep.py [n_queries] [budget]        (host cores only; no GPU needed)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake  # noqa: E402
from bio_ik_amd.workload import make_queries  # noqa: E402
from oracle import orc, ref  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    budget = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    pr2 = pr2_like()
    cases = [("C2 right_arm", ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")])),
             ("C3 all (2 tips + sec)", ProblemTemplate(pr2, "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), MinimalDisplacementGoal()])),
             ("C4 snake31 (+ sec)", ProblemTemplate(snake(31), "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()]))]
    for name, t in cases:
        o = orc.Oracle(t)
        for kind in ("global", "tracking"):
            seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=0xB101C, kind=kind)
            r = ref.Reference(t, abi.default_solve_params(mode="bio2_memetic", random_seed=1), release=True)
            r.solve_batch(seeds[:2], params[:2], 4)  # builds the solver and its random tables outside the timing
            t0 = time.perf_counter()
            _, _, suc, steps = r.solve_batch(seeds, params, budget)
            dt = time.perf_counter() - t0
            print("%-22s %-8s: %8.0f solves/s on one host thread  success %.4f  mean steps %.1f  (%d queries, <= %d steps, %.2f s)" %
                  (name, kind, suc.sum() / dt, suc.mean(), steps.mean(), n, budget, dt))


if __name__ == "__main__":
    main()
