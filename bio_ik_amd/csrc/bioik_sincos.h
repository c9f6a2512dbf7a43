// bioik_sincos.h — one sincos for every side of the boundary.
//
// The joint frames of revolute joints need sin/cos of the half angle (reference src/forward_kinematics.h:89-112
// calls libm).  libm on the host and the device math library disagree in the last ulp, and bio2_memetic amplifies
// last-ulp differences into different search trajectories (its line search divides by a second difference of
// fitness values taken 1e-7 apart, src/ik_evolution_2.cpp:498-506).  To make "same inputs -> same results" hold
// bit for bit between the gfx950 kernels and the CPU oracle, both evaluate THIS function: a two-term reduction by pi/2
// (pi/2 = P1 + P2, each step ONE fused multiply-add, exact under cancellation) followed by the fdlibm minimax kernels as plain
// Horner chains, written with *, rint and EXPLICIT fused multiply-adds (compiler contraction is off everywhere), so any IEEE-754
// double implementation produces identical bits.  Error <= 1.56 ulp for |x| <= 1e5 (joint half angles are a few radians):
// tests/test_oracle_frame.py.  32 instructions on gfx950 against 45 for the three-term reduction with a tail carried through the
// kernels (1.02 ulp) that it replaces: +3 % chip-wide step rate (profiles/r02_ab_sincos.log).
// Polynomial coefficients: fdlibm k_sin.c, k_cos.c (Sun Microsystems, 1993).
#pragma once

#ifndef BIOIK_SINCOS_FN
#define BIOIK_SINCOS_FN inline
#endif

BIOIK_SINCOS_FN void bioik_sincos(double x, double* sn, double* cs) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double invpio2 = 6.36619772367581382433e-01;
    const double P1 = 1.57079632679489655800e+00, P2 = 6.12323399573676603587e-17;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double fn = __builtin_rint(x * invpio2);
    double r = __builtin_fma(-fn, P1, x);
    r = __builtin_fma(-fn, P2, r);
    const double z = r * r;
    const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2), S1);
    const double s = __builtin_fma(z * r, ps, r);
    const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
    const double c = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    const int q = ((int)fn) & 3;
    const double so = (q & 1) ? c : s;
    const double co = (q & 1) ? s : c;
#if defined(BIOIK_SINCOS_SIGN_BRANCHES)
    *sn = q >= 2 ? -so : so;
    *cs = (q == 1 || q == 2) ? -co : co;
#else
    // quadrants 2, 3 negate the sine, quadrants 1, 2 the cosine: bit 1 of q (of q + 1) moved onto the sign bit -- a shift, an and, an
    // exclusive or on the high word instead of a comparison and a select per value (negation IS the sign flip, for every double)
    unsigned long long sb, cb;
    __builtin_memcpy(&sb, &so, 8);
    __builtin_memcpy(&cb, &co, 8);
    sb ^= (unsigned long long)(((unsigned)q << 30) & 0x80000000u) << 32;
    cb ^= (unsigned long long)((((unsigned)q + 1u) << 30) & 0x80000000u) << 32;
    __builtin_memcpy(sn, &sb, 8);
    __builtin_memcpy(cs, &cb, 8);
#endif
}
