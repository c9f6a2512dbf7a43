// bioik_acos.h — one acos and one atan2 for every side of the boundary (like bioik_sincos.h).
//
// The goal costs and the success test of the reference call libm's acos (tf2Acos behind tf2::Vector3::angle / Quaternion::angleShortestPath:
// include/bio_ik/goal_types.h:183-212 LookAtGoal ... :646-712 ConeGoal, src/problem.cpp:259-341) and atan2 (KDL::Rotation::GetRot behind the twist test).
// libm on the host and the device math library disagree in the last ulp, and an angle that enters a fitness value enters the line search's second
// differences (src/ik_evolution_2.cpp:498-506): whole solves with a LookAt / Cone goal then part.  Rounds 1 - 5 left those solves out of the bit-for-bit GPU
// suites; since round 6 the kernels and the CPU checker's "device arithmetic" mode evaluate THESE functions: the fdlibm algorithms (e_acos.c, s_atan.c,
// e_atan2.c; Sun Microsystems, 1993: rational minimax approximations, < 1 ulp) written with +, -, *, / and sqrt alone -- every one of them correctly rounded
// on either side (the device's double-precision sqrt and division are the IEEE ones: tools/micro/sqrt_check.hip, ten million arguments, 0 differences) and
// compiler contraction off everywhere -- so any IEEE-754 double implementation produces identical bits.  Held against mpmath / long double:
// tests/test_arith_headers.py (acos: error <= 1 ulp; atan2: <= 1.5 ulp -- the rounding of the quotient y / x counts double where the angle falls into the
// binade below the quotient's; libm's is 0.8 ulp.  The angle is compared with thresholds of 1e-5: what matters is that both sides compute the same one).
#pragma once

#ifndef BIOIK_ACOS_FN
#define BIOIK_ACOS_FN inline
#endif

// acos(x) for x in [-1, 1]; NaN for a NaN and for |x| > 1 (the callers clamp first, as tf2Acos does)
BIOIK_ACOS_FN double bioik_acos(double x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    if (!(x == x)) return x;
    const double ax = __builtin_fabs(x);
    if (ax >= 1.0) {
        if (ax > 1.0) return (x - x) / (x - x);  // NaN
        return x > 0.0 ? 0.0 : pi + 2.0 * pio2_lo;
    }
    if (ax < 0.5) {
        if (ax <= 6.938893903907228e-18) return pio2_hi + pio2_lo;  // |x| <= 2^-57
        const double z = x * x;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (x < 0.0) {
        const double z = (1.0 + x) * 0.5;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double s = __builtin_sqrt(z);
        const double r = p / q;
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    const double z = (1.0 - x) * 0.5;
    const double s = __builtin_sqrt(z);
    unsigned long long sb;
    __builtin_memcpy(&sb, &s, 8);
    sb &= 0xffffffff00000000ull;  // s with its low word cleared: df * df is exact
    double df;
    __builtin_memcpy(&df, &sb, 8);
    const double c = (z - df * df) / (s + df);
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double r = p / q;
    const double w = r * s + c;
    return 2.0 * (df + w);
}

// atan(x), any x (s_atan.c)
BIOIK_ACOS_FN double bioik_atan(double x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02, aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    if (!(x == x)) return x;
    const bool neg = __builtin_signbit(x);
    double ax = __builtin_fabs(x);
    if (ax >= 7.378697629483821e19) {  // 2^66
        const double z = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
        return neg ? -z : z;
    }
    int id = -1;
    double t = x;
    if (ax < 0.4375) {
        if (ax < 1.862645149230957e-09) return x;  // 2^-29
    } else {
        if (ax < 1.1875) {
            if (ax < 0.6875) id = 0, t = (2.0 * ax - 1.0) / (2.0 + ax);
            else id = 1, t = (ax - 1.0) / (ax + 1.0);
        } else {
            if (ax < 2.4375) id = 2, t = (ax - 1.5) / (1.0 + 1.5 * ax);
            else id = 3, t = -1.0 / ax;
        }
    }
    const double z = t * t;
    const double w = z * z;
    const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return t - t * (s1 + s2);
    // (selects, not a table: an indexed array would live in scratch memory on the device)
    const double hi = id == 0 ? 4.63647609000806093515e-01 : id == 1 ? 7.85398163397448278999e-01 : id == 2 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00;
    const double lo = id == 0 ? 2.26987774529616870924e-17 : id == 1 ? 3.06161699786838301793e-17 : id == 2 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17;
    const double r = hi - ((t * (s1 + s2) - lo) - t);
    return neg ? -r : r;
}

// atan2(y, x) (e_atan2.c)
BIOIK_ACOS_FN double bioik_atan2(double y, double x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double pi = 3.1415926535897931160E+00, pi_o_2 = 1.5707963267948965580E+00, pi_o_4 = 7.8539816339744827900E-01, pi_lo = 1.2246467991473531772E-16;
    if (!(x == x) || !(y == y)) return x + y;
    if (x == 1.0) return bioik_atan(y);
    const bool ny = __builtin_signbit(y), nx = __builtin_signbit(x);
    const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
    if (ay == 0.0) return nx ? (ny ? -pi : pi) : y;  // y = +-0: +-0 for x >= +0, +-pi for x <= -0
    if (ax == 0.0) return ny ? -pi_o_2 : pi_o_2;
    const double inf = __builtin_inf();
    if (ax == inf) {
        if (ay == inf) return nx ? (ny ? -3.0 * pi_o_4 : 3.0 * pi_o_4) : (ny ? -pi_o_4 : pi_o_4);
        return nx ? (ny ? -pi : pi) : (ny ? -0.0 : 0.0);
    }
    if (ay == inf) return ny ? -pi_o_2 : pi_o_2;
    // |y / x| beyond 2^60: pi / 2; x < 0 and |y / x| below 2^-60: 0 -- decided on the exponents, as e_atan2.c does
    unsigned long long by, bx;
    __builtin_memcpy(&by, &ay, 8);
    __builtin_memcpy(&bx, &ax, 8);
    const int k = (int)(by >> 52) - (int)(bx >> 52);
    double z;
    if (k > 60) z = pi_o_2 + 0.5 * pi_lo;
    else if (nx && k < -60) z = 0.0;
    else z = bioik_atan(__builtin_fabs(y / x));
    if (!nx) return ny ? -z : z;
    return ny ? (z - pi_lo) - pi : pi - (z - pi_lo);
}
