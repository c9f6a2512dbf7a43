"""The arithmetic BOTH sides of the boundary include -- bio_ik_amd/csrc/bioik_sincos.h and bioik_fused.h -- held against an independent reference.

The kernels are bit-identical to the CPU checker only in the checker's "device arithmetic" mode, and in that mode the checker includes these two
headers itself: a defect in them would be invisible to every bit-parity test.  Here each function is evaluated where it runs in production (on the
device through `bioik_eval_arith`; without a GPU: the same sources in the host simulator) for a large set of arguments and compared with

  * `mpmath` at 50 digits (a sample; tools/arith_reference.py is the generating script: nothing is stored, the references are computed at test time),
  * NumPy long double (64-bit mantissa, the bulk), whose own agreement with mpmath is asserted first,

with the error bounds asserted in ulps (sincos: the 1.56 ulp its header documents, SURVEY.md section 8(c) function level) or in units of
eps x the sum of the magnitudes of the terms (the fused dot / cross / Hamilton products: each result is a sum of products rounded a few times).
The sparse forms of a revolute joint (bioik_device.h: revolute_apply) must equal the general form wherever the struck-out constants are zeros."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import arith_reference as ar  # noqa: E402

from bio_ik_amd import solver  # noqa: E402

EPS = 2.0 ** -53


def _check_sincos(ev, n):
    rng = np.random.default_rng(20260927)
    xs = [rng.uniform(-4.0, 4.0, n), rng.uniform(-100.0, 100.0, n // 4), rng.uniform(-1e5, 1e5, n // 4), ar.sincos_special_arguments()]
    # Error model of the two-term reduction by pi/2 = P1 + P2 (106 bits): the reduced argument is off by at most |fn| x 1.5e-33 (the next term of pi/2
    # that is not carried), which is nothing next to a value of order one but IS visible, in ulps, in the component that vanishes when x sits within
    # 1e-16 of a multiple of pi/2 (cos of fl(pi/2): 6e-17).  So: error <= 1.6 ulp of the value + 1e-32 (|x| + 1) absolute, and <= 1.6 ulp outright on
    # the random arguments (bioik_sincos.h documents <= 1.56 ulp for |x| <= 1e5).
    worst = 0.0
    for i, x in enumerate(xs):
        out = ev(0, x)
        rs, rc = ar.sincos_longdouble(x)
        for got, ref in ((out[:, 0], rs), (out[:, 1], rc)):
            ulp = np.spacing(np.abs(ref.astype(np.float64)))
            err = np.abs(got.astype(np.longdouble) - ref)
            assert np.all(err <= 1.6 * ulp + 1e-32 * (np.abs(x) + 1.0)), "sincos: %.3g ulp beyond the reduction's own error" % float(np.max(err / ulp))
            if i < 3:
                worst = max(worst, float(np.max(err / ulp)))
        assert np.all(np.abs(out[:, 0] ** 2 + out[:, 1] ** 2 - 1.0) < 8 * EPS)
    assert worst <= 1.6, "sincos: %.3f ulp on random arguments" % worst
    # the long double reference itself against mpmath (50 digits) on a sample
    xm = np.concatenate([x[:400] for x in xs])
    ms, mc = ar.sincos_mpmath(xm)
    ls, lc = ar.sincos_longdouble(xm)
    assert float(np.max(np.abs(ls - ms))) < 2e-19 and float(np.max(np.abs(lc - mc))) < 2e-19
    out = ev(0, xm)
    ulp_s, ulp_c = np.spacing(np.abs(ms.astype(np.float64))), np.spacing(np.abs(mc.astype(np.float64)))
    assert np.all(np.abs(out[:, 0].astype(np.longdouble) - ms) <= 1.6 * ulp_s + 1e-32 * (np.abs(xm) + 1.0))
    assert np.all(np.abs(out[:, 1].astype(np.longdouble) - mc) <= 1.6 * ulp_c + 1e-32 * (np.abs(xm) + 1.0))
    return worst


def _check_fused(ev, n):
    rng = np.random.default_rng(7)
    scale = lambda shape: 10.0 ** rng.uniform(-3, 3, shape)  # noqa: E731
    for op, width, ref_fn, rounds in ((1, 7, ar.qrot_longdouble, 8), (2, 8, ar.qmul_longdouble, 5), (3, 6, ar.dot3_longdouble, 4), (4, 8, ar.dot4_longdouble, 5)):
        x = rng.normal(size=(n, width)) * scale((n, width))
        if op == 1:  # (rotations by unit quaternions as well: what the chain walk feeds it)
            x[: n // 2, :4] /= np.linalg.norm(x[: n // 2, :4], axis=1, keepdims=True)
        got = ev(op, x)
        ref, mag = ref_fn(x)
        err = np.abs(got.astype(np.longdouble) - ref)
        assert np.all(err <= rounds * EPS * mag), "op %d: %.3g eps x magnitude" % (op, float(np.max(err / (EPS * mag))))
    # exact cases: the products of small integers are exact in every form
    xi = rng.integers(-50, 50, size=(1000, 8)).astype(np.float64)
    assert np.array_equal(ev(2, xi), ar.qmul_longdouble(xi)[0].astype(np.float64))
    assert np.array_equal(ev(4, xi)[:, 0], ar.dot4_longdouble(xi)[0].astype(np.float64)[:, 0])


def _check_revolute(ev, n):
    rng = np.random.default_rng(11)
    for pos_kind in (1, 2, 3, 4):      # BIOIK_POS_ZERO, _X, _Y, _Z
        for rot_kind in (1, 2, 3):     # BIOIK_ROT_X, _Y, _Z
            x = np.zeros((n, 22))
            x[:, 0:3] = rng.normal(size=(n, 3))
            q = rng.normal(size=(n, 4))
            x[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
            x[:, 7] = rng.uniform(-3.2, 3.2, n)
            if pos_kind >= 2:
                x[:, 8 + pos_kind - 2] = rng.normal(size=n)  # cpos on one axis
            x[:, 14] = 1.0                                      # ca = (0, 0, 0, 1): unrotated constant frame
            x[:, 15 + rot_kind - 1] = rng.choice([-1.0, 1.0, 0.5], size=n)  # cb = (v e_i, 0)
            x[:, 19], x[:, 20] = pos_kind, rot_kind
            out = ev(5, x)
            assert np.all(out[:, :7] == out[:, 7:]), "sparse form (%d, %d) differs from the general form" % (pos_kind, rot_kind)
            # and the general form against the long double composition f o (cpos, cs ca + sn cb)
            ref, mag = ar.revolute_longdouble(x)
            assert np.all(np.abs(out[:, :7].astype(np.longdouble) - ref) <= 24 * EPS * mag)


def _check_acos_atan2(ev, n):
    """bioik_acos.h: the fdlibm algorithms.  acos held to 1 ulp, atan2 to 1.5 ulp against long double (itself held against mpmath at 50 digits on a sample), on
    random arguments, on the algorithm's branch points and next to +-1 where acos is steepest (the angle of two nearly parallel directions: ConeGoal, LookAtGoal)."""
    rng = np.random.default_rng(20261001)
    xs = np.concatenate([rng.uniform(-1.0, 1.0, n), 1.0 - 10.0 ** rng.uniform(-16, 0, n // 4), -1.0 + 10.0 ** rng.uniform(-16, 0, n // 4), ar.acos_special_arguments()])
    got = ev(6, xs)[:, 0]
    ref = ar.acos_longdouble(xs)
    ulp = np.spacing(np.abs(ref.astype(np.float64)))
    err = np.abs(got.astype(np.longdouble) - ref)
    worst_acos = float(np.max(err / ulp))
    assert worst_acos <= 1.0, "acos: %.3f ulp" % worst_acos
    assert got[xs == 1.0].max() == 0.0 and np.all(got[xs == -1.0] == np.pi) and np.all(got[xs == 0.0] == np.pi / 2)
    special = ev(6, np.array([np.nan, 1.0000000001, -2.0]))[:, 0]
    assert np.all(np.isnan(special))  # (the callers clamp; a NaN stays a NaN, as through tf2Acos)
    yx = np.concatenate([rng.normal(size=(n, 2)) * 10.0 ** rng.uniform(-3, 3, (n, 2)), np.stack([np.abs(rng.normal(size=n // 2)) * 1e-4, rng.uniform(-1.0, 1.0, n // 2)], axis=1),
                         ar.atan2_special_arguments()])
    got2 = ev(7, yx)[:, 0]
    ref2 = ar.atan2_longdouble(yx)
    ulp2 = np.spacing(np.abs(ref2.astype(np.float64)))
    err2 = np.abs(got2.astype(np.longdouble) - ref2)
    worst_atan2 = float(np.max(err2 / np.maximum(ulp2, 5e-324)))
    # (atan of the ROUNDED quotient y / x: the quotient's half ulp counts double where the angle falls into the binade below the quotient's, e.g. 0.2522 -> 0.24698)
    assert worst_atan2 <= 1.5, "atan2: %.3f ulp" % worst_atan2
    edge = ev(7, np.array([[0.0, 1.0], [0.0, -1.0], [-0.0, -1.0], [1.0, 0.0], [-1.0, 0.0], [np.nan, 1.0], [1.0, np.inf], [1.0, -np.inf], [np.inf, np.inf], [-np.inf, -np.inf]]))[:, 0]
    assert edge[0] == 0.0 and edge[1] == np.pi and edge[2] == -np.pi and edge[3] == np.pi / 2 and edge[4] == -np.pi / 2 and np.isnan(edge[5])
    assert edge[6] == 0.0 and edge[7] == np.pi and edge[8] == np.pi / 4 and edge[9] == -3 * np.pi / 4
    # the long double references themselves against mpmath (50 digits) on a sample
    xm = np.concatenate([xs[:300], ar.acos_special_arguments()])
    assert float(np.max(np.abs(ar.acos_longdouble(xm) - ar.acos_mpmath(xm)))) < 4e-19
    ym = np.concatenate([yx[:300], ar.atan2_special_arguments()[::7]])
    assert float(np.max(np.abs(ar.atan2_longdouble(ym) - ar.atan2_mpmath(ym)))) < 4e-19
    return worst_acos, worst_atan2


def test_shared_arithmetic_headers_in_the_host_simulator(hostsim_lib):
    """-m "not gpu": bioik_sincos.h / bioik_fused.h / bioik_acos.h as g++ compiles them (the bits the CPU checker's device-arithmetic mode computes)"""
    ev = lambda op, x: solver.eval_arith(op, x, lib=hostsim_lib)  # noqa: E731
    _check_sincos(ev, 40000)
    _check_fused(ev, 20000)
    _check_revolute(ev, 2000)
    _check_acos_atan2(ev, 40000)


@pytest.mark.gpu
def test_shared_arithmetic_headers_on_the_device():
    """the same on gfx950, a million arguments per function"""
    ev = lambda op, x: solver.eval_arith(op, x)  # noqa: E731
    worst = _check_sincos(ev, 1000000)
    _check_fused(ev, 1000000)
    _check_revolute(ev, 50000)
    wa, wt = _check_acos_atan2(ev, 1000000)
    print("sincos worst error %.3f ulp, acos %.3f ulp, atan2 %.3f ulp" % (worst, wa, wt))


@pytest.mark.gpu
def test_device_and_host_simulator_agree_bit_for_bit(hostsim_lib):
    """what makes bit parity between the kernels and the CPU checker possible at all: hipcc for gfx950 and g++ for x86-64 produce the same bits from these headers"""
    rng = np.random.default_rng(3)
    x = rng.uniform(-50.0, 50.0, 200000)
    assert np.array_equal(solver.eval_arith(0, x), solver.eval_arith(0, x, lib=hostsim_lib))
    for op, w in ((1, 7), (2, 8), (3, 6), (4, 8)):
        y = rng.normal(size=(100000, w))
        assert np.array_equal(solver.eval_arith(op, y), solver.eval_arith(op, y, lib=hostsim_lib))
    xa = np.concatenate([rng.uniform(-1.0, 1.0, 200000), 1.0 - 10.0 ** rng.uniform(-16, 0, 50000), ar.acos_special_arguments()])
    assert np.array_equal(solver.eval_arith(6, xa), solver.eval_arith(6, xa, lib=hostsim_lib))
    yx = np.concatenate([rng.normal(size=(200000, 2)) * 10.0 ** rng.uniform(-3, 3, (200000, 2)), ar.atan2_special_arguments()])
    assert np.array_equal(solver.eval_arith(7, yx), solver.eval_arith(7, yx, lib=hostsim_lib))
