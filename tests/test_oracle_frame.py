"""L1 math of the oracle pinned by the reference's OWN unit tests (test/utest.cpp) and by NumPy."""
import numpy as np

from oracle import orc


def random_frame(rng):
    # test/utest.cpp:52-60
    q = rng.uniform(-1, 1, 4)
    q /= np.linalg.norm(q)
    return np.concatenate([rng.uniform(-1, 1, 3), q])


def test_utest_change():
    """reference test/utest.cpp:63-81: change(c, b, concat(b, a)) == concat(c, a) to 1e-3 (here also to 1e-12)."""
    rng = np.random.default_rng(0)
    for _ in range(10000):
        fa, fb, fc = random_frame(rng), random_frame(rng), random_frame(rng)
        fx = orc.frame_concat(fb, fa)
        fy = orc.frame_concat(fc, fa)
        fz = orc.frame_change(fc, fb, fx)
        assert np.allclose(fy, fz, atol=1e-3, rtol=0)
        assert np.allclose(fy, fz, atol=1e-12, rtol=0)


def test_utest_linear_int_distribution():
    """reference test/utest.cpp:83-111: histogram of linear_int_distribution(8) ~ (8-i) to 1e-3."""
    n, iters = 8, 1000000
    v = orc.linear_int_distribution_hist(0, n, iters)
    r = np.array([n - i for i in range(n)], dtype=float)
    v /= v.sum()
    r /= r.sum()
    assert np.allclose(v, r, atol=0.001)


def test_quat_ops_against_rotation_matrices():
    from np_fk import quat_to_rot64
    rng = np.random.default_rng(1)
    for _ in range(200):
        a, b = random_frame(rng), random_frame(rng)
        v = rng.uniform(-2, 2, 3)
        Ra, Rb = quat_to_rot64(a[3:]), quat_to_rot64(b[3:])
        assert np.allclose(orc.quat_mul_vec(a[3:], v), Ra @ v, atol=1e-14)
        assert np.allclose(quat_to_rot64(orc.quat_mul_quat(a[3:], b[3:])), Ra @ Rb, atol=1e-14)
        c = orc.frame_concat(a, b)
        assert np.allclose(c[:3], a[:3] + Ra @ b[:3], atol=1e-14)
        assert np.allclose(quat_to_rot64(c[3:]), Ra @ Rb, atol=1e-14)
        inv = orc.frame_invert(a)
        assert np.allclose(orc.frame_concat(a, inv), [0, 0, 0, 0, 0, 0, 1], atol=1e-14)


def test_quat_mul_vec_short_circuits():
    # reference frame.h:122-126: identity quaternion or zero vector return v unchanged (bitwise)
    v = np.array([0.1, -0.2, 0.3])
    assert (orc.quat_mul_vec([0, 0, 0, 1], v) == v).all()
    assert (orc.quat_mul_vec([0.5, 0.5, 0.5, 0.5], [0, 0, 0]) == 0).all()


def test_normalize_fast_is_one_newton_step():
    # reference frame.h:231-238
    q = np.array([0.1, 0.2, 0.3, 0.9])
    q = q / np.linalg.norm(q) * 1.01
    f = (3.0 - q @ q) * 0.5
    assert np.allclose(orc.normalize_fast(q), q * f, atol=1e-16)
    assert abs(np.linalg.norm(orc.normalize_fast(q)) - 1) < 2e-4


def test_frame_twist_small_rotation():
    # reference frame.h:240-259: twist of b relative to a = (translation, axis*angle) in a's frame
    rng = np.random.default_rng(2)
    a = random_frame(rng)
    axis = np.array([0.0, 0.6, 0.8])
    ang = 0.3
    d = np.concatenate([[0.01, -0.02, 0.03], axis * np.sin(ang / 2), [np.cos(ang / 2)]])
    b = orc.frame_concat(a, d)
    tw = orc.frame_twist(a, b)
    assert np.allclose(tw[:3], d[:3], atol=1e-14)
    assert np.allclose(tw[3:], axis * ang, atol=1e-12)


def test_pose_twist_matches_rotation_vector():
    """KDL twist used by the dtwist success test (problem.cpp:316-323): translation and rotation vector of the tip
    relative to the goal, expressed in the goal frame."""
    rng = np.random.default_rng(3)
    from np_fk import quat_to_rot64
    for _ in range(100):
        goal = random_frame(rng)
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(1e-4, 3.0)
        dp = rng.uniform(-0.1, 0.1, 3)
        d = np.concatenate([dp, axis * np.sin(ang / 2), [np.cos(ang / 2)]])
        tip = orc.frame_concat(goal, d)
        tw = orc.pose_twist(goal, tip)
        assert np.allclose(tw[:3], dp, atol=1e-13)
        assert np.allclose(tw[3:], axis * ang, atol=1e-9)
    # identical frames -> exactly zero twist (KDL identity branch)
    assert np.allclose(orc.pose_twist(goal, goal), 0, atol=1e-15)


def test_shared_sincos_accuracy():
    """bioik_sincos (bio_ik_amd/csrc/bioik_sincos.h, evaluated by the kernels and by the oracle's device-arithmetic mode) against
    long-double sin / cos: <= 1.6 ulp over joint-value ranges and far beyond, including arguments next to multiples of pi/2"""
    from oracle import orc
    rng = np.random.default_rng(0)
    for scale in (1.0, 4.0, 100.0, 1e5):
        x = np.concatenate([rng.uniform(-scale, scale, 100000), np.arange(-8, 9) * np.pi / 2 + rng.normal(size=17) * 1e-9, np.arange(-8, 9) * np.pi / 4])
        s, c = orc.shared_sincos(x)
        xl = x.astype(np.longdouble)
        for got, ref in ((s, np.sin(xl)), (c, np.cos(xl))):
            sp = np.maximum(np.spacing(np.abs(ref.astype(np.float64))), np.finfo(float).tiny).astype(np.longdouble)
            assert np.abs((got.astype(np.longdouble) - ref) / sp).max() <= 1.6
        assert np.abs(s * s + c * c - 1.0).max() < 5e-16
