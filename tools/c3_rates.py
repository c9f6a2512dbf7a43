"""success rate of the two-armed problem (C3) and the snake (C4) at the step budgets the result-level tests use"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from bio_ik_amd import AvoidJointLimitsGoal, MinimalDisplacementGoal, PoseGoal, ProblemTemplate, abi, pr2_like, snake
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
t = ProblemTemplate(pr2_like(), "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), MinimalDisplacementGoal()])
h = HipSolver(t, device=0)
for seed in (1, 2, 3):
    seeds, params, _ = make_queries(t, h.active_variables, h.fk_genes, 4096, seed=seed)
    for steps in (64, 128):
        r = h.solve_batch(abi.default_solve_params(population=128, max_steps=steps, random_seed=seed), seeds, params)
        print("c3 seed %d steps %d: success %.4f" % (seed, steps, r[2].mean()), flush=True)
