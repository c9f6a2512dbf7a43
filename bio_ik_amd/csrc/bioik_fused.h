// bioik_fused.h — the fused-multiply-add forms of the hot arithmetic primitives, one definition for every side of the
// boundary (like bioik_sincos.h).
//
// The reference computes its frame algebra with separate multiplies and adds (include/bio_ik/frame.h:108-172).  On gfx950
// a v_fma_f64 costs the same issue slot as a v_mul_f64 or a v_add_f64, so writing the quaternion rotation, the Hamilton
// product and the dot products with explicit FMAs removes ~40 % of the FP64 instructions of a joint transform (and rounds
// once instead of twice: the results are at least as accurate).  Compilers disagree on WHICH multiply-add pairs to contract,
// so contraction is switched off everywhere (-ffp-contract=off) and the fused forms are spelled out here; the gfx950
// kernels always use them; the CPU checker of the test suite uses them in its "device arithmetic" mode and
// keeps the reference's unfused expressions in mode 0, where it is pinned bit-for-bit against the reference's own code.
#pragma once

#ifndef BIOIK_FUSED_FN
#define BIOIK_FUSED_FN inline
#endif
#define BK_FMA(a, b, c) __builtin_fma((a), (b), (c))

BIOIK_FUSED_FN double bk_dot3(double ax, double ay, double az, double bx, double by, double bz) { return BK_FMA(ax, bx, BK_FMA(ay, by, az * bz)); }
BIOIK_FUSED_FN double bk_dot4(double ax, double ay, double az, double aw, double bx, double by, double bz, double bw) {
    return BK_FMA(ax, bx, BK_FMA(ay, by, BK_FMA(az, bz, aw * bw)));
}
// v rotated by the unit quaternion q:  v + 2 (q.w t + q.xyz x t),  t = q.xyz x v     (frame.h:108-149)
BIOIK_FUSED_FN void bk_qrot(double qx, double qy, double qz, double qw, double vx, double vy, double vz, double& ox, double& oy, double& oz) {
    const double tx = BK_FMA(qy, vz, -(qz * vy));
    const double ty = BK_FMA(qz, vx, -(qx * vz));
    const double tz = BK_FMA(qx, vy, -(qy * vx));
    const double rx = BK_FMA(qw, tx, BK_FMA(qy, tz, -(qz * ty)));
    const double ry = BK_FMA(qw, ty, BK_FMA(qz, tx, -(qx * tz)));
    const double rz = BK_FMA(qw, tz, BK_FMA(qx, ty, -(qy * tx)));
    ox = BK_FMA(2.0, rx, vx);
    oy = BK_FMA(2.0, ry, vy);
    oz = BK_FMA(2.0, rz, vz);
}
// Hamilton product p (x) q   (frame.h:151-172)
BIOIK_FUSED_FN void bk_qmul(double px, double py, double pz, double pw, double qx, double qy, double qz, double qw, double& ox, double& oy, double& oz,
                            double& ow) {
    ox = BK_FMA(pw, qx, BK_FMA(px, qw, BK_FMA(py, qz, -(pz * qy))));
    oy = BK_FMA(pw, qy, BK_FMA(py, qw, BK_FMA(pz, qx, -(px * qz))));
    oz = BK_FMA(pw, qz, BK_FMA(pz, qw, BK_FMA(px, qy, -(py * qx))));
    ow = BK_FMA(pw, qw, -BK_FMA(px, qx, BK_FMA(py, qy, pz * qz)));
}
