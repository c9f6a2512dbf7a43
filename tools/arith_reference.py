"""Independent references for the shared arithmetic headers (bio_ik_amd/csrc/bioik_sincos.h, bioik_fused.h, bioik_acos.h): NumPy long double (x87 80-bit: 64-bit
mantissa) for bulk comparisons and mpmath (50 digits) to validate the long double values themselves.  Used by tests/test_arith_headers.py at test time
(nothing is stored); `python tools/arith_reference.py` prints the agreement of the two references.  Written from the mathematical definitions
(frame.h:108-172 of the reference for the quaternion algebra), not from the headers under test."""
import numpy as np

LD = np.longdouble


def sincos_special_arguments():
    k = np.arange(-64, 65, dtype=np.float64)
    base = np.concatenate([k * (np.pi / 4), k * (np.pi / 2)])
    near = np.concatenate([base, np.nextafter(base, np.inf), np.nextafter(base, -np.inf), base + 1e-9, base - 1e-9])
    return np.concatenate([near, [0.0, -0.0, 1e-300, -1e-300, 1e-8, 0.5, 1.0, 2.0, 3.0, 1e4 + 0.1]])


def sincos_longdouble(x):
    xl = np.asarray(x, dtype=np.float64).astype(LD)
    return np.sin(xl), np.cos(xl)


def sincos_mpmath(x):
    import mpmath
    mpmath.mp.dps = 50
    s = np.array([LD(mpmath.nstr(mpmath.sin(mpmath.mpf(float(v))), 30)) for v in x], dtype=LD)
    c = np.array([LD(mpmath.nstr(mpmath.cos(mpmath.mpf(float(v))), 30)) for v in x], dtype=LD)
    return s, c


def acos_special_arguments():
    """the branch points of the fdlibm algorithm (0.5, 1, tiny arguments), both signs, and their neighbours"""
    base = np.array([0.0, 2.0 ** -58, 2.0 ** -57, 2.0 ** -56, 1e-300, 1e-10, 0.25, 0.4999999, 0.5, 0.5000001, 0.75, 0.9, 0.99, 0.999999, 1.0 - 2.0 ** -52, 1.0 - 2.0 ** -53, 1.0])
    near = np.concatenate([base, np.nextafter(base, 2.0), np.nextafter(base, -2.0)])
    near = near[np.abs(near) <= 1.0]
    return np.concatenate([near, -near])


def acos_longdouble(x):
    return np.arccos(np.asarray(x, dtype=np.float64).astype(LD))


def acos_mpmath(x):
    import mpmath
    mpmath.mp.dps = 50
    return np.array([LD(mpmath.nstr(mpmath.acos(mpmath.mpf(float(v))), 30)) for v in x], dtype=LD)


def atan2_special_arguments():
    """the reduction thresholds of atan (0.4375, 0.6875, 1.1875, 2.4375) as ratios, the axes, both signs of everything"""
    r = np.array([0.0, 1e-300, 2.0 ** -30, 2.0 ** -29, 0.4374999, 0.4375, 0.6875, 1.0, 1.1875, 2.4375, 1e10, 2.0 ** 61, 1e300])
    r = np.concatenate([r, np.nextafter(r, np.inf), np.nextafter(r, -np.inf)])
    r = r[r >= 0.0]
    ys, xs = [], []
    for sy in (1.0, -1.0):
        for sx in (1.0, -1.0):
            ys.append(sy * r), xs.append(np.full(r.shape, sx))          # y / x = the ratio
            ys.append(np.full(r.shape, sy)), xs.append(sx * np.maximum(r, 1e-300))  # x / y = the ratio
    return np.stack([np.concatenate(ys), np.concatenate(xs)], axis=1)


def atan2_longdouble(yx):
    yx = np.asarray(yx, dtype=np.float64).astype(LD)
    return np.arctan2(yx[:, 0], yx[:, 1])


def atan2_mpmath(yx):
    import mpmath
    mpmath.mp.dps = 50
    return np.array([LD(mpmath.nstr(mpmath.atan2(mpmath.mpf(float(y)), mpmath.mpf(float(x))), 30)) for y, x in yx], dtype=LD)


def _cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)


def qrot_longdouble(x):
    """v rotated by q (not necessarily of unit length, as frame.h:108-149 computes it): v + 2 (w t + u x t), t = u x v, q = (u, w); and the magnitude of its terms"""
    x = np.asarray(x, dtype=np.float64).astype(LD)
    u, w, v = x[:, 0:3], x[:, 3:4], x[:, 4:7]
    t = _cross(u, v)
    r = v + 2 * (w * t + _cross(u, t))
    au, av = np.abs(u), np.abs(v)
    at = _cross(au, av) + 2 * np.stack([au[:, 2] * av[:, 1], au[:, 0] * av[:, 2], au[:, 1] * av[:, 0]], axis=1)  # |u| x |v| with every term positive
    mag = av + 2 * (np.abs(w) * at + _cross(au, at) + 2 * np.stack([au[:, 2] * at[:, 1], au[:, 0] * at[:, 2], au[:, 1] * at[:, 0]], axis=1))
    return r, mag


def qmul_longdouble(x):
    """Hamilton product p (x) q (frame.h:151-172), components x y z w"""
    x = np.asarray(x, dtype=np.float64).astype(LD)
    px, py, pz, pw, qx, qy, qz, qw = [x[:, i] for i in range(8)]
    terms = [[pw * qx, px * qw, py * qz, -pz * qy], [pw * qy, py * qw, pz * qx, -px * qz], [pw * qz, pz * qw, px * qy, -py * qx], [pw * qw, -px * qx, -py * qy, -pz * qz]]
    r = np.stack([sum(t) for t in terms], axis=1)
    mag = np.stack([sum(np.abs(v) for v in t) for t in terms], axis=1)
    return r, mag


def dot3_longdouble(x):
    x = np.asarray(x, dtype=np.float64).astype(LD)
    p = x[:, 0:3] * x[:, 3:6]
    return p.sum(axis=1, keepdims=True), np.abs(p).sum(axis=1, keepdims=True)


def dot4_longdouble(x):
    x = np.asarray(x, dtype=np.float64).astype(LD)
    p = x[:, 0:4] * x[:, 4:8]
    return p.sum(axis=1, keepdims=True), np.abs(p).sum(axis=1, keepdims=True)


def revolute_longdouble(x):
    """frame (p, q) o (cpos, cos(h) ca + sin(h) cb): p' = p + q cpos q^-1, q' = q (x) (cos(h) ca + sin(h) cb)   (forward_kinematics.h:89-112, :331-354)"""
    x = np.asarray(x, dtype=np.float64).astype(LD)
    p, q, h, cpos, ca, cb = x[:, 0:3], x[:, 3:7], x[:, 7:8], x[:, 8:11], x[:, 11:15], x[:, 15:19]
    lq = np.cos(h) * ca + np.sin(h) * cb
    rp, mp = qrot_longdouble(np.concatenate([q, cpos], axis=1).astype(np.float64))
    rq, mq = qmul_longdouble(np.concatenate([q, lq.astype(np.float64)], axis=1).astype(np.float64))
    return np.concatenate([p + rp, rq], axis=1), np.concatenate([np.abs(p) + mp, mq + 1e-300], axis=1)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-1e5, 1e5, 2000), sincos_special_arguments()])
    a, b = sincos_longdouble(xs), sincos_mpmath(xs)
    print("long double vs mpmath (50 digits), %d arguments: max |sin| diff %.3g, max |cos| diff %.3g" % (len(xs), float(np.max(np.abs(a[0] - b[0]))), float(np.max(np.abs(a[1] - b[1])))))
