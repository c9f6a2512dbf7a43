"""Distribution of step() counts over the bench workload (C2, 4096 queries): success rate as a function of the step budget."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
h = HipSolver(t, device=0)
seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 4096, seed=0xB101C)
p = abi.default_solve_params(population=128, max_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 256, random_seed=1)
sol, fit, suc, steps = h.solve_batch(p, seeds, params)
print("budget %d: success %.4f, mean steps %.2f" % (p.max_steps, suc.mean(), steps.mean()))
for b in (8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 96, 128, 192, 256):
    print("  solved within %3d steps: %.4f" % (b, ((suc == 1) & (steps <= b)).mean()))
