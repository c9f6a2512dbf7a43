"""L2 kinematics of the oracle: exact FK vs an independent NumPy rotation-matrix FK, analytic Jacobian vs finite
differences of the exact FK, mutation approximator vs exact FK (the idea of reference src/ik_test.cpp:92-128)."""
import numpy as np
import pytest

from conftest import random_configuration
from np_fk import fk_all, quat_to_rot64


@pytest.mark.parametrize("key", ["c2", "c3", "c4"])
def test_fk_matches_independent_numpy_fk(key, oracles, templates):
    o, m = oracles[key], templates[key].model
    rng = np.random.default_rng(10)
    for _ in range(10):
        v = random_configuration(m, rng)
        tips, glob = o.fk(v, want_global=True)
        R, p = fk_all(m, v)
        for t, link in enumerate(o.tip_links):
            assert np.allclose(tips[0, t, :3], np.asarray(p[link], dtype=float), atol=1e-13)
            assert np.allclose(quat_to_rot64(tips[0, t, 3:]), np.asarray(R[link], dtype=float), atol=1e-13)
            assert abs(np.linalg.norm(tips[0, t, 3:]) - 1) < 1e-14
        # every scheduled link (ancestors of tips) too
        for link in o.tip_links:
            l = link
            while l >= 0:
                assert np.allclose(glob[0, l, :3], np.asarray(p[l], dtype=float), atol=1e-13)
                l = m.link_parent[l]


def test_fk_zero_configuration_known_answer(oracles):
    # PR2-like right arm at zero: links stretched along +x (SURVEY.md Appendix C offsets)
    o = oracles["c2"]
    tips = o.fk(np.zeros(o.V))
    assert np.allclose(tips[0, 0], [-0.05 + 0.1 + 0.4 + 0.321, -0.188, 0.051 + 0.739675, 0, 0, 0, 1], atol=1e-15)
    o4 = oracles["c4"]
    assert np.allclose(o4.fk(np.zeros(o4.V))[0, 0], [3.1, 0, 0, 0, 0, 0, 1], atol=1e-15)


def test_fk_mimic_and_prismatic():
    from bio_ik_amd import PoseGoal, ProblemTemplate, RobotModel
    from oracle import orc
    m = RobotModel("mimic")
    m.add_link("base")
    m.add_link("a", "base", "ja", "revolute", xyz=(0, 0, 0.1), axis=(0, 0, 1), lower=-1, upper=1, velocity=1)
    m.add_link("b", "a", "jb", "revolute", xyz=(0.2, 0, 0), axis=(0, 1, 0), lower=-2, upper=2, velocity=1, mimic=("ja", 2.0, 0.1))
    m.add_link("c", "b", "jc", "prismatic", xyz=(0.3, 0, 0), axis=(1, 0, 0), lower=0, upper=0.5, velocity=1)
    m.add_group("g", joints=["ja", "jb", "jc"], tips=["c"])
    t = ProblemTemplate(m, "g", [PoseGoal("c")])
    o = orc.Oracle(t)
    assert list(o.active_variables) == [0, 2]  # mimic joint jb is not an active variable (problem.cpp:202)
    v = np.array([0.3, 123.0, 0.25])           # the mimic variable's own value is overwritten (forward_kinematics.h:230-246)
    tips = o.fk(v)
    v2 = v.copy()
    v2[1] = 0.3 * 2.0 + 0.1
    R, p = fk_all(m, v2)
    assert np.allclose(tips[0, 0, :3], np.asarray(p[3], dtype=float), atol=1e-14)
    # Jacobian column of ja includes the mimic joint scaled by the mimic factor (forward_kinematics.h:624-630)
    jac = o.jacobian(v, v[[0, 2]])
    eps = 1e-6
    g = v[[0, 2]].copy()
    fp = o.fk_genes(v, g + [eps, 0])[0, 0]
    fm = o.fk_genes(v, g - [eps, 0])[0, 0]
    Rt = quat_to_rot64(tips[0, 0, 3:])
    dp_local = Rt.T @ ((fp[:3] - fm[:3]) / (2 * eps))
    assert np.allclose(jac[:3, 0], dp_local, atol=1e-8)


@pytest.mark.parametrize("key", ["c2", "c3", "c4"])
def test_jacobian_matches_finite_differences(key, oracles, templates):
    """forward_kinematics.h:600-730: rows [v; omega] per tip in TIP-LOCAL coordinates."""
    o, m = oracles[key], templates[key].model
    rng = np.random.default_rng(11)
    seed = random_configuration(m, rng)
    base = seed[o.active_variables]
    jac = o.jacobian(seed, base)
    tips0 = o.fk_genes(seed, base)[0]
    eps = 1e-6
    for c in range(o.D):
        d = np.zeros(o.D)
        d[c] = eps
        fp, fm = o.fk_genes(seed, base + d)[0], o.fk_genes(seed, base - d)[0]
        for t in range(o.T):
            Rt = quat_to_rot64(tips0[t, 3:])
            v_local = Rt.T @ ((fp[t, :3] - fm[t, :3]) / (2 * eps))
            dR = (quat_to_rot64(fp[t, 3:]) - quat_to_rot64(fm[t, 3:])) / (2 * eps)
            W = Rt.T @ dR  # skew(omega_local)
            w_local = np.array([W[2, 1], W[0, 2], W[1, 0]])
            assert np.allclose(jac[6 * t:6 * t + 3, c], v_local, atol=2e-8)
            assert np.allclose(jac[6 * t + 3:6 * t + 6, c], w_local, atol=2e-8)


@pytest.mark.parametrize("key", ["c2", "c3", "c4"])
def test_mutation_approximator_is_first_order(key, oracles, templates):
    """forward_kinematics.h:802-930, 1175-1233: exact at the base, error O(delta^2) (ik_test.cpp measures the same)."""
    o, m = oracles[key], templates[key].model
    rng = np.random.default_rng(12)
    seed = random_configuration(m, rng)
    base = seed[o.active_variables]
    tips, deltas, mask = o.approximator(seed, base)
    assert np.allclose(tips, o.fk_genes(seed, base)[0], atol=0)
    assert np.array_equal(o.approx_eval(seed, base, base)[0], tips)
    errs = []
    for scale in (1e-2, 1e-3):
        g = base + scale * rng.normal(size=o.D)
        lin = o.approx_eval(seed, base, g)[0]
        ex = o.fk_genes(seed, g)[0]
        # quaternion sign: exact FK is continuous here so signs agree
        errs.append(np.abs(lin - ex).max())
    assert errs[0] < 5e-3 * max(1.0, o.D / 7)
    assert errs[1] < errs[0] * 0.03  # quadratic: 100x smaller for a 10x smaller step (loose factor)
    # linear phenotypes are the plain sum base + sum_c delta[t][c]*(g_c - base_c), unnormalised (quirk kept)
    g = base + 0.05 * rng.normal(size=o.D)
    lin = o.approx_eval(seed, base, g)[0]
    manual = tips + np.einsum("tck,c->tk", deltas, g - base)
    assert np.allclose(lin, manual, atol=1e-15)


def test_approximator_mask_two_tips(oracles, templates):
    """forward_kinematics.h:907-929: a variable contributes to a tip iff its delta frame is non-zero."""
    o = oracles["c3"]
    m = templates["c3"].model
    rng = np.random.default_rng(13)
    seed = random_configuration(m, rng)
    _, deltas, mask = o.approximator(seed, seed[o.active_variables])
    names = [m.variable_names[v] for v in o.active_variables]
    for c, n in enumerate(names):
        if n.startswith("r_"):
            assert mask[0, c] == 1 and mask[1, c] == 0 and np.all(deltas[1, c] == 0)
        elif n.startswith("l_"):
            assert mask[0, c] == 0 and mask[1, c] == 1 and np.all(deltas[0, c] == 0)
        else:  # torso moves both
            assert mask[0, c] == 1 and mask[1, c] == 1


def test_robot_info(oracles, templates):
    """robot_info.h:70-106: continuous joints become unbounded for clipping but keep min/max/span."""
    o, m = oracles["c2"], templates["c2"].model
    info = o.robot_info()
    v = m.variable_index("r_forearm_roll_joint")
    assert info[v, 0] == -np.finfo(float).max and info[v, 1] == np.finfo(float).max
    assert np.isclose(info[v, 2], 2 * np.pi) and np.isclose(info[v, 3], -np.pi) and np.isclose(info[v, 4], np.pi)
    v = m.variable_index("r_elbow_flex_joint")
    assert info[v, 0] == -2.3213 and info[v, 1] == 0.0 and np.isclose(info[v, 2], 2.3213)
    assert np.isclose(info[v, 5], 1 / 3.3)
    # minimal displacement weights (problem.cpp:207-225): rcp(vmax)/sum
    w = o.velocity_weights()
    rcp = np.array([1.0 / m.var_max_velocity[i] for i in o.active_variables])
    assert np.allclose(w, rcp / rcp.sum(), atol=1e-16)
