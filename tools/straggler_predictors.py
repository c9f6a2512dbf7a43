"""What predicts how many steps a query needs?  CPU only (oracle, counter RNG): rank correlation of the step count with the step count under
another random seed, the goal's distance from the shoulder, the initial error, the fitness after K steps; and a coarse list-scheduling simulation of
an isolated call with the hand-over ordered by fitness / by the true remaining steps.  usage: python tools/straggler_predictors.py"""
import numpy as np, sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from bio_ik_amd import PoseGoal, ProblemTemplate, abi, pr2_like
from bio_ik_amd.workload import make_queries
from oracle import orc
from scipy.stats import spearmanr
t = ProblemTemplate(pr2_like(), "right_arm", [PoseGoal("r_wrist_roll_link")])
o = orc.Oracle(t)
orc.set_trig_mode(1)
n=2048
seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=0xB101C)
p = abi.default_solve_params(population=128, max_steps=64, random_seed=1)
t0=time.time(); sol, fit, suc, steps = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=32); print('oracle %.1f s'%(time.time()-t0), suc.mean(), steps.mean())
p2 = abi.default_solve_params(population=128, max_steps=64, random_seed=7)
sol2, fit2, suc2, steps2 = o.solve_batch(p2, orc.RNG_COUNTER, seeds, params, n_threads=32)
print('steps vs steps under another random seed: spearman %.3f' % spearmanr(steps, steps2).correlation)
hard = steps>=32; print('P(hard again | hard) = %.2f, base rate %.3f' % ((steps2[hard]>=32).mean(), (steps2>=32).mean()))
pos = params[:, :3]
r = np.linalg.norm(pos - np.array([0.0,-0.188,0.79]), axis=1)
print('radius from shoulder: spearman %.3f' % spearmanr(steps, r).correlation)
# initial fitness (seed pose error)
tips0 = o.fk(seeds)
f0 = np.linalg.norm(tips0[:,0,:3]-pos, axis=1)
print('initial position error: spearman %.3f' % spearmanr(steps, f0).correlation)
for K in (1,2,4,8):
    pk = abi.default_solve_params(population=128, max_steps=K, random_seed=1)
    sk = o.solve_batch(pk, orc.RNG_COUNTER, seeds, params, n_threads=32)
    m = steps > K
    print('fitness after %d steps (unsolved only): spearman %.3f' % (K, spearmanr(steps[m], sk[1][m]).correlation))

import heapq
def simulate(rem, order, slots, dt_full=0.39, dt_lone=0.10):
    # list scheduling in the given order; every running job advances one step at a time, step time depends on how many run
    pending = list(order); running = []  # (remaining steps)
    t = 0.0; active = []
    # time-stepped simulation with variable step duration (all running jobs step together: coarse but adequate)
    rem_run = []
    while pending or rem_run:
        while pending and len(rem_run) < slots:
            rem_run.append(rem[pending.pop(0)])
        load = min(1.0, len(rem_run) / slots)
        t += dt_lone + (dt_full - dt_lone) * load
        rem_run = [r - 1 for r in rem_run if r > 1]
    return t
slots = 768
print('one launch, natural order: %.2f ms' % simulate(steps, list(range(n)), slots))
print('one launch, true steps descending (unknowable): %.2f ms' % simulate(steps, list(np.argsort(-steps)), slots))
for K in (2, 4, 8):
    pk = abi.default_solve_params(population=128, max_steps=K, random_seed=1)
    sk = o.solve_batch(pk, orc.RNG_COUNTER, seeds, params, n_threads=32)
    alive = np.nonzero(steps > K)[0]
    rem = steps - K
    t1 = simulate(np.minimum(steps, K), list(range(n)), slots)
    nat = simulate(rem, list(alive), slots)
    byfit = simulate(rem, list(alive[np.argsort(-sk[1][alive])]), slots)
    best = simulate(rem, list(alive[np.argsort(-rem[alive])]), slots)
    print('hand-over after %d steps: first launch %.2f ms + second: natural %.2f, by fitness (worst first) %.2f, by true remaining steps %.2f ms' % (K, t1, nat, byfit, best))
