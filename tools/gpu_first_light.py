"""First-light script for a GPU box: function-level + step-level parity against the oracle and a rough timing sweep."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bio_ik_amd import *
from bio_ik_amd import abi
from bio_ik_amd.solver import HipSolver
from bio_ik_amd.workload import make_queries
from oracle import orc
from conftest import random_configuration

pr2 = pr2_like(); sn = snake(31)
T = {"c2": ProblemTemplate(pr2, "right_arm", [PoseGoal("r_wrist_roll_link")]),
     "c3": ProblemTemplate(pr2, "all", [PoseGoal("r_wrist_roll_link"), PoseGoal("l_wrist_roll_link"), MinimalDisplacementGoal()]),
     "c4": ProblemTemplate(sn, "snake", [PoseGoal("tip"), AvoidJointLimitsGoal()])}
rng = np.random.default_rng(0)
orc.set_trig_mode(1)
if os.environ.get("BIOIK_LIB"):
    from bio_ik_amd import solver as _s
    _s._lib = _s.load_library(os.environ["BIOIK_LIB"])
H = {}
for k, t in T.items():
    o = orc.Oracle(t); h = HipSolver(t); H[k] = (o, h)
    seed = random_configuration(t.model, rng)
    genes = random_configuration(t.model, rng, 1000)[:, o.active_variables]
    print(k, "fk maxdiff", np.abs(o.fk_genes(seed, genes) - h.fk_genes(seed, genes)).max(), flush=True)
    par = rng.normal(size=o.P)
    pa, sa = o.fitness(abi.FK_EXACT, seed, par, genes); pb, sb = h.fitness(abi.FK_EXACT, seed, par, genes)
    print(k, "fit exact rel", (np.abs(pa - pb) / np.abs(pa)).max(), np.abs(sa - sb).max())
    base = genes[0]; g2 = base + 0.01 * rng.normal(size=(500, o.D))
    pa, sa = o.fitness(abi.FK_LINEAR, seed, par, g2, base); pb, sb = h.fitness(abi.FK_LINEAR, seed, par, g2, base)
    print(k, "fit linear rel", (np.abs(pa - pb) / np.abs(pa)).max(), np.abs(sa - sb).max())
    ta, da, ma = o.approximator(seed, base); tb, db = h.approximator(seed, base)
    print(k, "approx", np.abs(ta - tb).max(), np.abs(da - db).max())
    parents = rng.normal(size=(2, 2, o.D)) * 0.1; parents[:, 0, :] = random_configuration(t.model, rng, 2)[:, o.active_variables]
    ga, gra = o.reproduce_counter(128, 12345, 1, 37, parents); gb, grb = h.reproduce(128, 12345, 1, 37, parents)
    print(k, "reproduce", np.abs(ga - gb).max(), np.abs(gra - grb).max())
    p = abi.default_solve_params()
    print(k, "check mismatches", (o.check(p, seed, par, genes) != h.check(p, seed, par, genes)).sum(), flush=True)

for k, fkm, pop in [("c2", -1, 16), ("c3", -1, 24), ("c2", abi.FK_EXACT, 16), ("c2", abi.FK_LINEAR, 16), ("c3", abi.FK_EXACT, 24), ("c4", abi.FK_LINEAR, 16), ("c2", abi.FK_EXACT, 128), ("c4", abi.FK_EXACT, 512)]:
    o, h = H[k]; t = T[k]; n = 16
    seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=7)
    for steps in (1, 2, 5):
        p = abi.default_solve_params(population=pop, fk_mode=fkm, max_steps=steps, random_seed=11) if fkm >= 0 else \
            abi.default_solve_params(population=pop, mode="bio2", max_steps=steps, random_seed=11)
        sa = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=8); sb = h.solve_batch(p, seeds, params)
        print(k, fkm, pop, "steps", steps, "sol maxdiff", np.abs(sa[0] - sb[0]).max(), "fit rel", (np.abs(sa[1] - sb[1]) / np.abs(sa[1])).max(),
              "suc", sa[2].sum(), sb[2].sum(), flush=True)

# timing sweep
for k, pop, n, ms in [("c2", 128, 4096, 100), ("c3", 128, 4096, 100), ("c4", 512, 4096, 100)]:
    o, h = H[k]; t = T[k]
    seeds, params, _ = make_queries(t, o.active_variables, h.fk_genes, n, seed=0xB101C)
    for thr in ("64", "128", "256"):
        os.environ["BIOIK_SOLVE_THREADS"] = thr
        p = abi.default_solve_params(population=pop, max_steps=ms, random_seed=1)
        h.solve_batch(p, seeds[:64], params[:64])
        t0 = time.time(); sol, fit, suc, steps = h.solve_batch(p, seeds, params); dt = time.time() - t0
        tips = o.fk(sol)
        perr = np.linalg.norm(tips[:, 0, :3] - params[:, :3], axis=1)
        print(k, "pop", pop, "threads", thr, "n", n, "time %.3fs" % dt, "solves/s %.0f" % (suc.sum() / dt), "success %.4f" % suc.mean(),
              "mean steps %.2f max %d" % (steps.mean(), steps.max()), "max pos err among successes %.2e" % perr[suc == 1].max(), flush=True)
    if k == "c2":
        t0 = time.time(); so = o.solve_batch(p, orc.RNG_COUNTER, seeds[:64], params[:64], n_threads=1); dt = time.time() - t0
        print("oracle 1 thread pop", pop, "64 queries: %.3fs -> %.1f solves/s, success %.3f mean steps %.2f" % (dt, so[2].sum() / dt, so[2].mean(), so[3].mean()), flush=True)
