"""The two arithmetic modes of the CPU oracle against each other (CPU only).

Mode 0 evaluates the reference's own expressions with libm sin / cos: it is the mode pinned bit for bit against the reference's
code (tests/test_oracle_vs_reference.py).  Mode 1 ("device arithmetic": bioik_sincos + the fused multiply-adds of
csrc/bioik_fused.h) is what the HIP kernels reproduce bit for bit (tests/test_gpu_parity.py).  This file closes the chain
reference == mode 0 ~ mode 1 == device at function level: the two modes may differ by a rounding per operation and no more."""
import numpy as np
import pytest

import parity_cases as pc
from bio_ik_amd import ProblemTemplate, abi
from conftest import gnarly_goals, mimic_robot, random_configuration
from oracle import orc

FRAME_TOL = 1e-14   # absolute, on frames of O(1) m / unit quaternions and on the approximator tables
FIT_RTOL = 1e-13    # relative, on fitness sums


def _cases(templates, gnarly):
    from bio_ik_amd import MinimalDisplacementGoal, PoseGoal, PositionGoal
    sec = MinimalDisplacementGoal(weight=0.5)
    sec.secondary_ = True
    m = mimic_robot()
    out = [(k, t, t.model) for k, t in templates.items()]
    out.append(("gnarly", ProblemTemplate(gnarly, "body", gnarly_goals()), gnarly))
    out.append(("mimic", ProblemTemplate(m, "arm", [PoseGoal("tool"), PositionGoal("finger_r_tip", weight=0.3), sec]), m))
    return out


def _both_modes(fn):
    with pc.oracle_arithmetic(0):
        a = fn()
    with pc.oracle_arithmetic(1):
        b = fn()
    return a, b


def test_device_arithmetic_mode_tracks_reference_arithmetic_mode(templates, gnarly):
    worst = {}
    for name, t, model in _cases(templates, gnarly):
        o = orc.Oracle(t)
        rng = np.random.default_rng(17)
        seed = random_configuration(model, rng)
        genes = random_configuration(model, rng, 400)[:, o.active_variables]
        par = rng.normal(size=o.P)
        # exact FK of every tip
        f0, f1 = _both_modes(lambda: o.fk_genes(seed, genes))
        assert not np.array_equal(f0, f1), "the two modes are expected to differ in the last bits (else this test compares nothing)"
        d_fk = np.abs(f0 - f1).max()
        assert d_fk <= FRAME_TOL, (name, d_fk)
        # goal fitness on exact phenotypes
        (p0, s0), (p1, s1) = _both_modes(lambda: o.fitness(abi.FK_EXACT, seed, par, genes))
        d_fit = (np.abs(p0 - p1) / np.maximum(np.abs(p0), 1e-300)).max()
        assert d_fit <= FIT_RTOL, (name, d_fit)
        assert np.all(np.abs(s0 - s1) <= FIT_RTOL * np.abs(s0))
        # mutation approximator tables and fitness on linearised phenotypes
        base = genes[0]
        (t0, dl0, m0), (t1, dl1, m1) = _both_modes(lambda: o.approximator(seed, base))
        assert np.array_equal(m0, m1)
        d_tab = max(np.abs(t0 - t1).max(), np.abs(dl0 - dl1).max())
        assert d_tab <= FRAME_TOL, (name, d_tab)
        near = base + 0.01 * rng.normal(size=(200, o.D))
        (p0, _), (p1, _) = _both_modes(lambda: o.fitness(abi.FK_LINEAR, seed, par, near, base))
        d_lin = (np.abs(p0 - p1) / np.maximum(np.abs(p0), 1e-300)).max()
        assert d_lin <= FIT_RTOL, (name, d_lin)
        worst[name] = (d_fk, d_fit, d_tab, d_lin)
    print("mode 1 vs mode 0 (max |d frame|, rel d fitness, |d table|, rel d linear fitness):", worst)


def test_modes_agree_on_reproduction_and_success_test(templates):
    """reproduction draws no trigonometry and no fused product: identical bits in both modes; the success test decides alike away
    from its thresholds"""
    t = templates["c2"]
    o = orc.Oracle(t)
    rng = np.random.default_rng(3)
    parents = rng.normal(size=(2, 2, o.D)) * 0.1
    parents[:, 0, :] = random_configuration(t.model, rng, 2)[:, o.active_variables]
    a, b = _both_modes(lambda: o.reproduce_counter(130, 0xC0FFEE, 1, 77, parents))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    seed = random_configuration(t.model, rng)
    genes = random_configuration(t.model, rng, 300)[:, o.active_variables]
    with pc.oracle_arithmetic(0):
        tips = o.fk_genes(seed, genes)
    p = abi.default_solve_params(dpos=0.05, drot=5.0, dtwist=-1.0)
    par = np.concatenate([tips[0, 0], [0.5]])
    near = genes[0] + rng.normal(size=(300, o.D)) * rng.choice([0.0, 1e-4, 1e-2, 0.3], size=(300, 1))
    c0, c1 = _both_modes(lambda: o.check(p, seed, par, near))
    assert np.array_equal(c0, c1) and c0.any() and not c0.all()


def test_kernel_bodies_against_reference_arithmetic(hostsim_lib, templates, gnarly):
    """the CPU twin of tests/test_gpu_parity_mode0.py: the kernel bodies (host simulator) against the oracle in mode 0, where the two
    sides share no arithmetic header — FK, fitness, approximator tables within 1e-12 / 1e-10 relative, children and success flags equal"""
    from bio_ik_amd.solver import HipSolver
    with pc.oracle_arithmetic(0):
        for name, t, model in _cases(templates, gnarly):
            pc.function_level(HipSolver(t, lib=hostsim_lib), orc.Oracle(t), model, np.random.default_rng(41), n=64, frame_tol=1e-12, fit_rtol=1e-10)


def test_whole_solves_are_equivalent_in_distribution_across_the_modes(templates):
    """Trajectories across the mode boundary: a last-bit difference in one fitness value flips a selection sooner or later (measured:
    within the first step -- eight generations and a memetic phase -- for every query of this batch), after which the two modes
    follow different but equally distributed random searches.  What can be asserted about whole solves is therefore statistical:
    the same success rate and the same step-count distribution (here: means within 10 %, 128 queries), and every success of either
    mode reproducing its goal pose under the reference-pinned arithmetic (mode 0)."""
    from bio_ik_amd.workload import make_queries
    t = templates["c2"]
    o = orc.Oracle(t)
    with pc.oracle_arithmetic(0):
        seeds, params, _ = make_queries(t, o.active_variables, o.fk_genes, 128, seed=5)
    p = abi.default_solve_params(population=64, max_steps=64, random_seed=3)
    a, b = _both_modes(lambda: o.solve_batch(p, orc.RNG_COUNTER, seeds, params, n_threads=8))
    assert not np.array_equal(a[0], b[0])
    assert abs(int(a[2].sum()) - int(b[2].sum())) <= 3 and a[2].mean() > 0.95
    assert abs(a[3].mean() - b[3].mean()) <= 0.1 * a[3].mean(), (a[3].mean(), b[3].mean())
    with pc.oracle_arithmetic(0):
        for sol, suc in ((a[0], a[2]), (b[0], b[2])):
            tips = o.fk(sol)
            for k in np.nonzero(suc)[0]:
                assert np.linalg.norm(tips[k, 0, :3] - params[k, :3]) < 1e-4
                assert 2 * np.arccos(min(1.0, abs(float(tips[k, 0, 3:] @ params[k, 3:7])))) < 1e-3
