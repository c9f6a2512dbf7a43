"""L3 solver of the oracle (reference src/ik_evolution_2.cpp, src/ik_parallel.h): operator-level properties and the
reference's own integration methodology, the FK->IK->FK round trip (README.md:404-447)."""
import numpy as np
import pytest

from bio_ik_amd import abi
from bio_ik_amd.workload import make_queries
from conftest import random_configuration
from oracle import orc


def test_reproduce_matches_formula(oracles, templates):
    """ik_evolution_2.cpp:263-300 restated with NumPy on top of the counter RNG."""
    o, t = oracles["c2"], templates["c2"]
    rng = np.random.default_rng(30)
    D = o.D
    parents = rng.normal(size=(2, 2, D)) * 0.1
    parents[:, 0, :] = random_configuration(t.model, rng, 2)[:, o.active_variables]
    info = o.robot_info()[o.active_variables]
    lam, key, sp, gen = 16, 12345, 1, 37
    genes, grads = o.reproduce_counter(lam, key, sp, gen, parents)
    c1 = (gen << 4) | (sp << 3) | 0
    for c in range(2, 2 + lam):
        k = orc.child_word(key, c1, c, 0) >> 28  # random word 0 of the child (its base): the top four bits
        rate = (1 << k) * (1.0 / (1 << 23))
        fmix = 0.2 if c % 2 == 0 else 0.0
        gf = float(c % 3)
        for g in range(D):
            z = orc.counter_child_gauss(key, c, g, c1)  # random word 1 + g
            m = parents[0, 1, g] * (1.0 - fmix) + parents[1, 1, g] * fmix
            x = parents[0, 0, g] + z * (rate * info[g, 2])
            x = x + m * gf
            x = min(max(x, info[g, 0]), info[g, 1])
            assert genes[c - 2, g] == pytest.approx(x, rel=1e-15, abs=1e-18)
            assert grads[c - 2, g] == pytest.approx(m * 0.7 + (x - parents[0, 0, g]) * 0.3, rel=1e-13, abs=1e-18)
    # clamping respects clip limits; continuous joints are unbounded (robot_info.h:82-90)
    assert np.all(genes >= info[:, 0]) and np.all(genes <= info[:, 1])


@pytest.mark.parametrize("rng_mode", [orc.RNG_REFERENCE, orc.RNG_COUNTER])
def test_round_trip_c2(oracles, templates, rng_mode):
    """FK -> IK -> FK on random valid configurations: every returned solution reproduces the goal pose."""
    o, t = oracles["c2"], templates["c2"]
    n = 40
    seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=5)
    p = abi.default_solve_params(population=16, fk_mode=abi.FK_LINEAR, max_steps=400, random_seed=3)
    sol, fit, suc, steps = o.solve_batch(p, rng_mode, seeds, params)
    assert suc.mean() >= 0.97
    tips = o.fk(sol)
    lo, hi = np.asarray(t.model.var_min), np.asarray(t.model.var_max)
    info = o.robot_info()
    for k in range(n):
        if not suc[k]:
            continue
        assert np.linalg.norm(tips[k, 0, :3] - params[k, :3]) < 1e-4   # north-star tolerance 1e-4 m
        q, qg = tips[k, 0, 3:], params[k, 3:7]
        ang = 2 * np.arccos(min(1.0, abs(q @ qg)))
        assert ang < 1e-3                                              # 1e-3 rad
        bounded = info[:, 1] != np.finfo(float).max
        assert np.all(sol[k][bounded] >= lo[bounded] - 1e-12) and np.all(sol[k][bounded] <= hi[bounded] + 1e-12)
        assert fit[k] < 1e-9
    # inactive variables are returned unchanged
    inactive = [v for v in range(o.V) if v not in list(o.active_variables)]
    assert np.array_equal(sol[:, inactive], seeds[:, inactive])


@pytest.mark.parametrize("key,pop,fk_mode,budget", [("c3", 128, abi.FK_LINEAR, 1500), ("c4", 16, abi.FK_LINEAR, 300),
                                                     ("c2", 128, abi.FK_EXACT, 300)])
def test_round_trip_other_configs(oracles, templates, key, pop, fk_mode, budget):
    # c3: the secondary MinimalDisplacementGoal (weight 1) slows convergence a lot with the reference population
    # (16 children: ~60 % success in 2000 steps); 128 children reach 100 % within ~800 steps.
    o, t = oracles[key], templates[key]
    n = 8
    seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, n, seed=6)
    p = abi.default_solve_params(population=pop, fk_mode=fk_mode, max_steps=budget, random_seed=4)
    sol, fit, suc, steps = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=4)
    assert suc.mean() >= 0.85
    tips = o.fk(sol)
    off = 0
    for gi, g in enumerate(t.goals):
        if g.opcode != abi.GOAL_POSE:
            continue
        ti = [gg for gg in t.goals if gg.link_name() is not None].index(g)
        for k in range(n):
            if suc[k]:
                assert np.linalg.norm(tips[k, ti, :3] - params[k, t.param_offsets[gi]:t.param_offsets[gi] + 3]) < 1e-4


def test_counter_solver_is_deterministic_and_thread_independent(oracles, templates):
    o, t = oracles["c2"], templates["c2"]
    seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 16, seed=7)
    p = abi.default_solve_params(population=32, fk_mode=abi.FK_LINEAR, max_steps=100, random_seed=11)
    a = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=1)
    b = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # query index selects the stream: shifting first_query_index changes trajectories
    c = o.solve_batch(p, orc.RNG_COUNTER, seeds, params, first_query_index=100)
    assert not np.array_equal(a[3], c[3]) or not np.array_equal(a[0], c[0])


def test_step_state_invariants(oracles, templates):
    """ik_evolution_2.cpp:604-645: species sorted by exact fitness, solution never gets worse, elites inside limits."""
    o, t = oracles["c3"], templates["c3"]
    seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 1, seed=8)
    p = abi.default_solve_params(population=16, fk_mode=abi.FK_LINEAR, max_steps=1)
    s = o.solver(p, orc.RNG_COUNTER, 99, seeds[0], params[0])
    info = o.robot_info()[o.active_variables]
    last = np.inf
    g0, f0, sol0, sf0 = s.state()
    assert np.array_equal(g0[:, :, 0, :], np.broadcast_to(seeds[0][o.active_variables], (2, 2, o.D)))  # :141-179 all = seed
    assert np.all(g0[:, :, 1, :] == 0)
    for _ in range(30):
        s.step()
        g, f, sol, sf = s.state()
        assert f[0] <= f[1]
        assert sf <= last
        last = sf
        assert np.all(g[:, :, 0, :] >= info[:, 0]) and np.all(g[:, :, 0, :] <= info[:, 1])
        prim, _ = o.fitness(abi.FK_EXACT, seeds[0], params[0], sol[o.active_variables])
        assert prim[0] == pytest.approx(sf, rel=1e-12, abs=1e-300)


def test_wrap_angles(oracles, templates):
    """kinematics_plugin.cpp:580-613: solution moved within +-pi of the seed, then into the limits."""
    o, t = oracles["c2"], templates["c2"]
    m = t.model
    seed = m.default_positions()
    v_roll = m.variable_index("r_forearm_roll_joint")   # continuous: limits [-pi, pi]
    v_pan = m.variable_index("r_shoulder_pan_joint")    # [-2.2854, 0.7146]
    seed[v_roll] = 0.5
    state = seed.copy()
    state[v_roll] = 0.5 + 2 * np.pi * 3 + 0.1           # three turns away -> comes back near the seed
    state[v_pan] = 0.9                                   # beyond the upper limit -> clamped
    out = o.wrap_angles(seed, state)
    assert out[v_roll] == pytest.approx(0.6, abs=1e-12)
    assert out[v_pan] == pytest.approx(0.7146)
    state[v_pan] = 0.5 - 2 * np.pi                      # a full turn below an admissible angle -> wrapped up by 2 pi
    seed[v_pan] = state[v_pan]
    out = o.wrap_angles(seed, state)
    assert out[v_pan] == pytest.approx(0.5, abs=1e-12)
