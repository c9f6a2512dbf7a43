// bioik_device.h — per-lane device functions of the MI355X bio2_memetic solver: rigid-transform algebra, the
// counter-based RNG, goal costs, the joint-program chain walk (exact FK), the linearised phenotype model and
// reproduction.  Every function is what ONE lane does for ONE individual; the kernels in bioik_kernels.h decide
// which individual a lane holds.  Written against bioik_platform.h only.
//
// Behavioural reference (what is computed, not how): TAMS-Group/bio_ik
//   include/bio_ik/frame.h:108-187, 231-238       quat_mul_vec, quat_mul_quat, concat, normalizeFast
//   src/forward_kinematics.h:78-139, 331-354       joint frames, exact FK over the link schedule
//   src/forward_kinematics.h:600-730, 802-930      analytic Jacobian, mutation approximator tables
//   src/forward_kinematics.h:1009-1058, 1175-1233  linearised phenotypes
//   include/bio_ik/goal_types.h:80-712             closed-form goal costs
//   src/problem.cpp:244-341                        computeGoalFitness, checkSolutionActiveVariables
//   src/ik_evolution_2.cpp:242-326                 reproduce
#pragma once
#include <type_traits>

#include "bioik_platform.h"


// goal opcodes / modes: numeric values of include/bioik_hip.h (kept in sync by a static_assert in bioik_hip.hip)
enum {
    G_POSITION = 0, G_ORIENTATION = 1, G_POSE = 2, G_LOOK_AT = 3, G_MAX_DISTANCE = 4, G_MIN_DISTANCE = 5, G_LINE = 6,
    G_PLANE = 7, G_AVOID_JOINT_LIMITS = 8, G_CENTER_JOINTS = 9, G_REGULARIZATION = 10, G_MINIMAL_DISPLACEMENT = 11,
    G_JOINT_VARIABLE = 12, G_SIDE = 13, G_DIRECTION = 14, G_CONE = 15
};
enum { FK_LINEAR = 0, FK_EXACT = 1 };

struct V3 {
    double x, y, z;
};
struct Q4 {
    double x, y, z, w;
};
struct F7 {
    V3 p;
    Q4 q;
};

BIOIK_DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
BIOIK_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
BIOIK_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
BIOIK_DEV V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
BIOIK_DEV double dot3(V3 a, V3 b) { return bk_dot3(a.x, a.y, a.z, b.x, b.y, b.z); }
BIOIK_DEV V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
BIOIK_DEV double len2(V3 a) { return dot3(a, a); }
BIOIK_DEV double dist2(V3 a, V3 b) { return len2(b - a); }
BIOIK_DEV V3 normalized3(V3 a) {  // tf2: v * (1 / length)
    const double inv = 1.0 / sqrt(len2(a));
    return V3{a.x * inv, a.y * inv, a.z * inv};
}
BIOIK_DEV double qdot(Q4 a, Q4 b) { return bk_dot4(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w); }
BIOIK_DEV Q4 qinv(Q4 q) { return Q4{-q.x, -q.y, -q.z, q.w}; }
// tf2Acos (tf2/LinearMath/Scalar.h): two comparisons that clamp the argument -- and let a NaN through, as the reference's does: acos(NaN) is NaN, and a goal that
// takes max(0, .) of its angle then costs nothing.  (fmin / fmax would make -1 of the NaN and pi of the angle: what separated the kernels from the oracle on the
// frames of a linear model evaluated at 1e308, profiles/r06_robot_fuzz_hostsim.log.)
BIOIK_DEV double clamped_acos(double x) {
    x = x < -1.0 ? -1.0 : x;
    x = x > 1.0 ? 1.0 : x;
    return bioik_acos(x);  // (the shared implementation, bioik_acos.h: the device library's and libm's differ in the last ulp)
}
BIOIK_DEV F7 f7_identity() { return F7{{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 1.0}}; }

// rotate v by unit quaternion q (frame.h:108-149; its identity/zero short-cuts are arithmetic no-ops); fused form
BIOIK_DEV V3 qrot(Q4 q, V3 v) {
    V3 o;
    bk_qrot(q.x, q.y, q.z, q.w, v.x, v.y, v.z, o.x, o.y, o.z);
    return o;
}
// Hamilton product (frame.h:151-172); fused form
BIOIK_DEV Q4 qmul(Q4 p, Q4 q) {
    Q4 o;
    bk_qmul(p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w, o.x, o.y, o.z, o.w);
    return o;
}
BIOIK_DEV F7 f7_concat(const F7& a, const F7& b) { return F7{a.p + qrot(a.q, b.p), qmul(a.q, b.q)}; }

// ---------------------------------------------------------------------------------------------------------
// A revolute joint applied to the running frame f:  f.p += f.q * cpos,  f.q = f.q (x) (cs * ca + sn * cb)
// (forward_kinematics.h:89-112, :331-354 with the fixed links folded into (cpos, ca), bioik_types.h).  The general form costs
// 21 + 8 + 16 FP64 instructions.  pos_kind / rot_kind (wavefront-uniform, from the problem compiler) name the constants that
// are exact zeros; a dropped term is `x * 0 + y`, which IS y, so every branch below returns the numbers of the general form
// (up to the sign of an exact zero).  Derivation: bk_qrot / bk_qmul of bioik_fused.h with the zero operands struck out, and
// `p + (2 r + 0)` written as the single rounding fma(2, r, p) (doubling is exact).
// ---------------------------------------------------------------------------------------------------------
template <int AXIS>  // v = cpos[AXIS] on axis AXIS, the other two components are zero
BIOIK_DEV void revolute_pos_axis(F7& f, double v) {
    const Q4 q = f.q;
    if constexpr (AXIS == 0) {
        const double ty = q.z * v, tz = -(q.y * v);
        const double rx = BK_FMA(q.y, tz, -(q.z * ty)), ry = BK_FMA(q.w, ty, -(q.x * tz)), rz = BK_FMA(q.w, tz, q.x * ty);
        f.p = V3{f.p.x + BK_FMA(2.0, rx, v), BK_FMA(2.0, ry, f.p.y), BK_FMA(2.0, rz, f.p.z)};
    } else if constexpr (AXIS == 1) {
        const double tx = -(q.z * v), tz = q.x * v;
        const double rx = BK_FMA(q.w, tx, q.y * tz), ry = BK_FMA(q.z, tx, -(q.x * tz)), rz = BK_FMA(q.w, tz, -(q.y * tx));
        f.p = V3{BK_FMA(2.0, rx, f.p.x), f.p.y + BK_FMA(2.0, ry, v), BK_FMA(2.0, rz, f.p.z)};
    } else {
        const double tx = q.y * v, ty = -(q.x * v);
        const double rx = BK_FMA(q.w, tx, -(q.z * ty)), ry = BK_FMA(q.w, ty, q.z * tx), rz = BK_FMA(q.x, ty, -(q.y * tx));
        f.p = V3{BK_FMA(2.0, rx, f.p.x), BK_FMA(2.0, ry, f.p.y), f.p.z + BK_FMA(2.0, rz, v)};
    }
}
template <int AXIS>  // local rotation (a e_AXIS, c): unrotated constant frame, joint axis on a coordinate axis; a = sn * cb[AXIS], c = cs
BIOIK_DEV void revolute_rot_axis(F7& f, double a, double c) {
    const Q4 p = f.q;
    if constexpr (AXIS == 0)
        f.q = Q4{BK_FMA(p.w, a, p.x * c), BK_FMA(p.y, c, p.z * a), BK_FMA(p.z, c, -(p.y * a)), BK_FMA(p.w, c, -(p.x * a))};
    else if constexpr (AXIS == 1)
        f.q = Q4{BK_FMA(p.x, c, -(p.z * a)), BK_FMA(p.w, a, p.y * c), BK_FMA(p.z, c, p.x * a), BK_FMA(p.w, c, -(p.y * a))};
    else
        f.q = Q4{BK_FMA(p.x, c, p.y * a), BK_FMA(p.y, c, -(p.x * a)), BK_FMA(p.w, a, p.z * c), BK_FMA(p.w, c, -(p.z * a))};
}
struct RevConst {  // the constants of one revolute op as the walk holds them (scalar registers)
    double cp0, cp1, cp2, ca0, ca1, ca2, ca3, cb0, cb1, cb2, cb3;
    int pos_kind, rot_kind;
};
// N individuals through the same joint: the branches are wavefront-uniform, the constants stay scalar operands
template <int N>
BIOIK_DEV void revolute_apply(F7 (&f)[N], const double (&sn)[N], const double (&cs)[N], const RevConst& k) {
    // position first: it reads the frame's rotation in front of the joint
    if (k.pos_kind == BIOIK_POS_GENERAL) {
#pragma unroll
        for (int j = 0; j < N; j++) f[j].p = f[j].p + qrot(f[j].q, v3(k.cp0, k.cp1, k.cp2));
    } else if (k.pos_kind == BIOIK_POS_X) {
#pragma unroll
        for (int j = 0; j < N; j++) revolute_pos_axis<0>(f[j], k.cp0);
    } else if (k.pos_kind == BIOIK_POS_Y) {
#pragma unroll
        for (int j = 0; j < N; j++) revolute_pos_axis<1>(f[j], k.cp1);
    } else if (k.pos_kind == BIOIK_POS_Z) {
#pragma unroll
        for (int j = 0; j < N; j++) revolute_pos_axis<2>(f[j], k.cp2);
    }  // BIOIK_POS_ZERO: the joint sits at its parent's origin
    if (k.rot_kind == BIOIK_ROT_GENERAL) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            const Q4 lq = Q4{BK_FMA(cs[j], k.ca0, sn[j] * k.cb0), BK_FMA(cs[j], k.ca1, sn[j] * k.cb1), BK_FMA(cs[j], k.ca2, sn[j] * k.cb2), BK_FMA(cs[j], k.ca3, sn[j] * k.cb3)};
            f[j].q = qmul(f[j].q, lq);
        }
    } else if (k.rot_kind == BIOIK_ROT_X) {
#pragma unroll
        for (int j = 0; j < N; j++) revolute_rot_axis<0>(f[j], sn[j] * k.cb0, cs[j]);
    } else if (k.rot_kind == BIOIK_ROT_Y) {
#pragma unroll
        for (int j = 0; j < N; j++) revolute_rot_axis<1>(f[j], sn[j] * k.cb1, cs[j]);
    } else {
#pragma unroll
        for (int j = 0; j < N; j++) revolute_rot_axis<2>(f[j], sn[j] * k.cb2, cs[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Counter-based RNG: Philox2x32-10 (Salmon et al., SC'11; Random123 constants) with integer-only
// post-processing, so that the device and the CPU restatement used by the tests produce bit-identical doubles.
// ---------------------------------------------------------------------------------------------------------
enum { RNG_REPRODUCE = 0, RNG_PRESELECT = 1, RNG_MEMETIC_SIGN = 2, RNG_WIPEOUT = 3, RNG_WIPEOUT_GENE = 4, RNG_POINT_RANDOM = 5 };
// Random words of child c in one generation -- the bulk of all draws, 1 + D words per child -- come from avalanche hashes of the counter instead of
// Philox (a Philox2x32-10 call is twenty 32-bit multiplies).  Round 6 (profiles/r06_issue_cost.log: on a SIMD that has several wavefronts to choose from, a
// 32-bit multiply takes the slot of an FP64 FMA, a VOP2 shift or xor half of one -- a gene's draw costs what its instructions COUNT):
//     stream        = mix32(key ^ ctr1 * 0x85EBCA77)                      wavefront-uniform (scalar unit), one per species and generation
//     base of child = mix32((c << 8) * 0x9E3779B1 + stream)               one FULL avalanche per child: children share nothing
//     word w >= 1   = mix1(base + w * 0x9E3779B1)                          one multiply per word: the words of ONE child are the lighter hash of a Weyl sequence
//                                                                          that starts at the child's own random point
// mix32 = MurmurHash3's 32-bit finaliser (two multiplies), mix1 = its first half (xor-shift 16, * 0x85EBCA6B, xor-shift 13).  The child's base is its word 0:
// its top four bits are the mutation-rate exponent; word 1 + g -> the Gaussian of gene g.  (Until round 5 every word was a mix32 of the counter: five more
// instructions per gene.)  Philox2x32-10 keeps the control draws (query key, pre-selection count, memetic sign, wipe-out).
BIOIK_DEV uint32_t rng_mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
BIOIK_DEV uint32_t rng_mix1(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    return h;
}
BIOIK_DEV uint32_t rng_child_stream(uint32_t key, uint32_t ctr1) { return rng_mix32(key ^ (ctr1 * 0x85EBCA77u)); }
BIOIK_DEV uint32_t rng_child_base(uint32_t stream, uint32_t child) { return rng_mix32((child << 8) * 0x9E3779B1u + stream); }  // word 0 of the child
BIOIK_DEV uint32_t rng_child_word(uint32_t base, uint32_t w) { return rng_mix1(base + w * 0x9E3779B1u); }                   // word w >= 1
BIOIK_DEV double rng_child_rate(uint32_t base) { return (double)(1u << (base >> 28)) * (1.0 / (double)(1 << 23)); }           // ik_evolution_2.cpp:283 with the counter's exponent

BIOIK_DEV void philox2x32_10(uint32_t key, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        if (r > 0) key += 0x9E3779B9u;
        uint64_t p = (uint64_t)0xD256D193u * (uint64_t)c0;
        uint32_t hi = (uint32_t)(p >> 32), lo = (uint32_t)p;
        c0 = hi ^ key ^ c1;
        c1 = lo;
    }
    o0 = c0;
    o1 = c1;
}
BIOIK_DEV uint32_t rng_ctr0(uint32_t child, uint32_t slot) { return (child << 8) | slot; }
BIOIK_DEV uint32_t rng_ctr1(uint32_t generation, uint32_t species, uint32_t purpose) { return (generation << 4) | (species << 3) | purpose; }
// ~N(0,1) from ONE 32-bit word: the sum of its four bytes (Irwin-Hall, n = 4: mean 510, variance 4 (256^2 - 1) / 12 = 21845 -- unit variance after the scaling,
// support +-3.451, a piecewise-cubic density, distribution function within 0.84 % of the normal one everywhere, kurtosis 2.70; tests/test_oracle_rng.py
// enumerates it).  Three instructions -- v_sad_u8 against zero with -510 as its addend, one exact conversion, one rounding -- where the Binomial(16) lattice
// with triangular jitter of rounds 1 - 5 took eight.
#define BIOIK_GAUSS_SCALE 0.006765875086793228 /* 1 / sqrt(21845) */
BIOIK_DEV double rng_gauss32(uint32_t x) { return (double)p_byte_sum(x, -510) * BIOIK_GAUSS_SCALE; }
BIOIK_DEV double rng_uniform(uint32_t x0, uint32_t x1) {
    uint64_t u = (((uint64_t)x0 << 32) | (uint64_t)x1) >> 11;
    return (double)u * (1.0 / 9007199254740992.0);
}
BIOIK_DEV uint32_t rng_query_key(uint64_t seed, uint64_t query, uint32_t island) {
    uint32_t o0, o1;
    philox2x32_10((uint32_t)seed, (uint32_t)query, (uint32_t)(seed >> 32) ^ (island * 0x9E3779B9u) ^ (uint32_t)(query >> 32), o0, o1);
    return o0;
}

// ---------------------------------------------------------------------------------------------------------
// per-query context: seed (Problem::initial_guess) and goal parameters, staged in LDS by the kernels
// ---------------------------------------------------------------------------------------------------------
struct QueryCtx {
    const double* seed;  // [V]
    const double* par;   // [P]
};

BIOIK_DEV F7 f7_load(const double* p) { return F7{{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}}; }
BIOIK_DEV void f7_store(double* p, const F7& f) {
    p[0] = f.p.x;
    p[1] = f.p.y;
    p[2] = f.p.z;
    p[3] = f.q.x;
    p[4] = f.q.y;
    p[5] = f.q.z;
    p[6] = f.q.w;
}

// One value per op of the joint program ("genotype in op order"), held in LDS.  A lane's own individual is a column
// of the [op][lane] array (p = base + lane, s = nthreads: consecutive lanes hit consecutive banks); a vector shared
// by the whole workgroup (an elite, a line-search point) is read with s = 1 (same address in every lane: broadcast).
struct XV {
    const double* p;
    int s;
    BIOIK_DEV double operator()(int k) const { return p[(size_t)k * s]; }
};

// ---------------------------------------------------------------------------------------------------------
// goal costs (goal_types.h); joint-set goals walk the active ops.
// ---------------------------------------------------------------------------------------------------------
// The link goals beyond position / orientation / pose: one out-of-line copy (sqrt, divisions, acos) instead of one per
// evaluation site.  P: the goal's numbers (at most 11, goal_types.h) where they lie in LDS -- the pointer crosses the call with its
// address space spelled out, and the 16 argument registers stay inside the 32 the calling convention passes without stack traffic.
BIOIK_CALL double goal_eval_link_rare(int type, const lds_f64* P, F7 fb) {
    switch (type) {
        case G_LOOK_AT: {  // :204-211
            V3 axis = qrot(fb.q, v3(P[0], P[1], P[2]));
            V3 target = v3(P[3], P[4], P[5]);
            return dist2(normalized3(target - fb.p), normalized3(axis));
        }
        case G_MAX_DISTANCE: {  // :235-240
            double d = fmax(0.0, sqrt(dist2(fb.p, v3(P[0], P[1], P[2]))) - P[3]);
            return d * d;
        }
        case G_MIN_DISTANCE: {  // :264-269
            double d = fmax(0.0, P[3] - sqrt(dist2(fb.p, v3(P[0], P[1], P[2]))));
            return d * d;
        }
        case G_LINE: {  // :293-297
            V3 position = v3(P[0], P[1], P[2]), direction = v3(P[3], P[4], P[5]);
            return dist2(position, fb.p - direction * dot3(direction, fb.p - position));
        }
        case G_PLANE: {  // :321-327
            V3 position = v3(P[0], P[1], P[2]), normal = v3(P[3], P[4], P[5]);
            double sd = dot3(fb.p - position, normal);
            return sd * sd;
        }
        case G_SIDE: {  // :606-613
            V3 v = qrot(fb.q, v3(P[0], P[1], P[2]));
            double f = fmax(0.0, dot3(v, v3(P[3], P[4], P[5])));
            return f * f;
        }
        case G_DIRECTION: {  // :637-643
            V3 v = qrot(fb.q, v3(P[0], P[1], P[2]));
            return dist2(v, v3(P[3], P[4], P[5]));
        }
        case G_CONE: {  // :700-711
            V3 v = qrot(fb.q, v3(P[4], P[5], P[6]));
            V3 dir = v3(P[7], P[8], P[9]);
            double ang = clamped_acos(dot3(v, dir) / sqrt(len2(v) * len2(dir)));
            double d = fmax(0.0, ang - P[10]);
            double w = P[3];
            return d * d + w * w * len2(v3(P[0], P[1], P[2]) - fb.p);
        }
    }
    return 0.0;
}
// The goals over the joint values (goal_types.h:387-498): one out-of-line copy; x and the seed are LDS pointers that cross the
// call with their address space spelled out.
// x: any "one value per op" accessor (an LDS vector / column, or a child computed where it is read)
template <class XA>
BIOIK_DEV double goal_eval_joint_set_x(ProbPtr pb, int type, int var_op, int var_seed, double p0, const XA& x, const lds_f64* seed) {
    // The reference's sums run over the ACTIVE VARIABLES -- the genes in their order.  That is the order of the ops wherever the ops meet the genes in it
    // (DevProblem::genes_follow_ops: every problem whose goals name no variable of their own); a JointVariableGoal puts its variable in front of the
    // chains' (problem.cpp:139-162), and the terms are then added in that order: the same terms, another rounding.
    const bool by_op = pb->genes_follow_ops != 0;
    const int cnt = by_op ? pb->n_ops : pb->D;
    switch (type) {
        case G_AVOID_JOINT_LIMITS: {  // :387-401
            double sum = 0.0;
            for (int i = 0; i < cnt; i++) {
                const int k = by_op ? i : pb->op_of_gene[i];
                if (pb->ops[k].gene >= 0 && !pb->ops[k].unbounded) {
                    double d = x(k) - (pb->ops[k].vmin + pb->ops[k].vmax) * 0.5;
                    d = fmax(0.0, fabs(d) * 2.0 - pb->ops[k].span * 0.5);
                    d *= pb->ops[k].vw;
                    sum += d * d;
                }
            }
            return sum;
        }
        case G_CENTER_JOINTS: {  // :412-425
            double sum = 0.0;
            for (int i = 0; i < cnt; i++) {
                const int k = by_op ? i : pb->op_of_gene[i];
                if (pb->ops[k].gene >= 0 && !pb->ops[k].unbounded) {
                    double d = x(k) - (pb->ops[k].vmin + pb->ops[k].vmax) * 0.5;
                    d *= pb->ops[k].vw;
                    sum += d * d;
                }
            }
            return sum;
        }
        case G_REGULARIZATION: {  // :435-444
            double sum = 0.0;
            for (int i = 0; i < cnt; i++) {
                const int k = by_op ? i : pb->op_of_gene[i];
                if (pb->ops[k].gene >= 0) {
                    double d = x(k) - seed[pb->ops[k].var];
                    sum += d * d;
                }
            }
            return sum;
        }
        case G_MINIMAL_DISPLACEMENT: {  // :455-465
            double sum = 0.0;
            for (int i = 0; i < cnt; i++) {
                const int k = by_op ? i : pb->op_of_gene[i];
                if (pb->ops[k].gene >= 0) {
                    double d = x(k) - seed[pb->ops[k].var];
                    d *= pb->ops[k].vw;
                    sum += d * d;
                }
            }
            return sum;
        }
        case G_JOINT_VARIABLE: {  // :494-498 + goal.h:70-77
            double v = var_op >= 0 ? x(var_op) : seed[var_seed];
            double d = p0 - v;
            return d * d;
        }
    }
    return 0.0;
}

// One op's term of the sums above (AvoidJointLimits, CenterJoints, Regularization, MinimalDisplacement): the operations of goal_eval_joint_set_x on the value
// xv of op k; 0 for an op the goal passes over -- `sum += 0` leaves a non-negative sum as it is.  The memetic phase's line search evaluates these goals on
// vectors that all of its lanes share, so lane k computes the term of op k once and every lane adds the terms up in their order (solve_body).
BIOIK_DEV bool joint_set_is_sum(int type) { return type == G_AVOID_JOINT_LIMITS || type == G_CENTER_JOINTS || type == G_REGULARIZATION || type == G_MINIMAL_DISPLACEMENT; }
template <class PB>
BIOIK_DEV double joint_set_term(PB pb, int type, int k, double xv, const double* seed) {
    if (pb->ops[k].gene < 0) return 0.0;
    switch (type) {
        case G_AVOID_JOINT_LIMITS: {
            if (pb->ops[k].unbounded) return 0.0;
            double d = xv - (pb->ops[k].vmin + pb->ops[k].vmax) * 0.5;
            d = fmax(0.0, fabs(d) * 2.0 - pb->ops[k].span * 0.5);
            d *= pb->ops[k].vw;
            return d * d;
        }
        case G_CENTER_JOINTS: {
            if (pb->ops[k].unbounded) return 0.0;
            double d = xv - (pb->ops[k].vmin + pb->ops[k].vmax) * 0.5;
            d *= pb->ops[k].vw;
            return d * d;
        }
        case G_REGULARIZATION: {
            const double d = xv - seed[pb->ops[k].var];
            return d * d;
        }
        case G_MINIMAL_DISPLACEMENT: {
            double d = xv - seed[pb->ops[k].var];
            d *= pb->ops[k].vw;
            return d * d;
        }
    }
    return 0.0;
}

// the same goals for N individuals at once (N independent dependency chains; per individual the operations and their order are those of
// goal_eval_joint_set_x): the pre-selection scores every child of a generation on its secondary goals, and with the children computed
// where they are read a gene is a hash, a Gaussian and a clip -- a long chain per gene that one individual at a time leaves exposed
// inside_mask (wavefront-uniform; AvoidJointLimitsGoal only): bit k = op k of EVERY individual in x lies strictly inside the half of its range that
// costs nothing (avoid_limits_surely_free) -- its term is `sum += 0`, which leaves a non-negative sum as it is, so the op is passed over without its
// value being computed at all
template <int N, class XA>
BIOIK_DEV void goal_eval_joint_set_xn(ProbPtr pb, int type, int var_op, int var_seed, double p0, const XA (&x)[N], const lds_f64* seed, double (&out)[N],
                                      uint64_t inside_mask = 0ull) {
    const int n_ops = pb->n_ops;
    double sum[N];
#pragma unroll
    for (int j = 0; j < N; j++) sum[j] = 0.0;
    switch (type) {
        case G_AVOID_JOINT_LIMITS:
            for (int k = 0; k < n_ops; k++)
                if (pb->ops[k].gene >= 0 && !pb->ops[k].unbounded && !((inside_mask >> k) & 1ull)) {
                    const double mid = (pb->ops[k].vmin + pb->ops[k].vmax) * 0.5, half_span = pb->ops[k].span * 0.5, vw = pb->ops[k].vw;
#pragma unroll
                    for (int j = 0; j < N; j++) {
                        double d = x[j](k) - mid;
                        d = fmax(0.0, fabs(d) * 2.0 - half_span);
                        d *= vw;
                        sum[j] += d * d;
                    }
                }
            break;
        case G_CENTER_JOINTS:
            for (int k = 0; k < n_ops; k++)
                if (pb->ops[k].gene >= 0 && !pb->ops[k].unbounded) {
                    const double mid = (pb->ops[k].vmin + pb->ops[k].vmax) * 0.5, vw = pb->ops[k].vw;
#pragma unroll
                    for (int j = 0; j < N; j++) {
                        double d = x[j](k) - mid;
                        d *= vw;
                        sum[j] += d * d;
                    }
                }
            break;
        case G_REGULARIZATION:
            for (int k = 0; k < n_ops; k++)
                if (pb->ops[k].gene >= 0) {
                    const double sv = seed[pb->ops[k].var];
#pragma unroll
                    for (int j = 0; j < N; j++) {
                        double d = x[j](k) - sv;
                        sum[j] += d * d;
                    }
                }
            break;
        case G_MINIMAL_DISPLACEMENT:
            for (int k = 0; k < n_ops; k++)
                if (pb->ops[k].gene >= 0) {
                    const double sv = seed[pb->ops[k].var], vw = pb->ops[k].vw;
#pragma unroll
                    for (int j = 0; j < N; j++) {
                        double d = x[j](k) - sv;
                        d *= vw;
                        sum[j] += d * d;
                    }
                }
            break;
        case G_JOINT_VARIABLE:
#pragma unroll
            for (int j = 0; j < N; j++) {
                const double v = var_op >= 0 ? x[j](var_op) : seed[var_seed];
                const double d = p0 - v;
                sum[j] = d * d;
            }
            break;
    }
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = sum[j];
}

struct LdsX {  // an op-indexed vector in LDS with its address space spelled out (crosses a call boundary)
    const lds_f64* p;
    int s;
    BIOIK_DEV double operator()(int k) const { return p[(size_t)k * s]; }
};
BIOIK_DEV double goal_eval_joint_set_inl(ProbPtr pb, int type, int var_op, int var_seed, double p0, const lds_f64* xp, int xs, const lds_f64* seed) {
    return goal_eval_joint_set_x(pb, type, var_op, var_seed, p0, LdsX{xp, xs}, seed);
}
BIOIK_CALL double goal_eval_joint_set(ProbPtr pb, int type, int var_op, int var_seed, double p0, const lds_f64* xp, int xs, const lds_f64* seed) {
    return goal_eval_joint_set_inl(pb, type, var_op, var_seed, p0, xp, xs, seed);
}
// PoseGoal::evaluate (goal_types.h:149-180); P: position, orientation, rotation scale
BIOIK_DEV double pose_goal_cost(const double* P, const F7& fb) {
    double e = dist2(fb.p, v3(P[0], P[1], P[2]));
    const Q4 d = Q4{P[3] - fb.q.x, P[4] - fb.q.y, P[5] - fb.q.z, P[6] - fb.q.w};
    const Q4 a = Q4{P[3] + fb.q.x, P[4] + fb.q.y, P[5] + fb.q.z, P[6] + fb.q.w};
    double rs = P[7];
    e += fmin(qdot(d, d), qdot(a, a)) * (rs * rs);
    return e;
}
// JS_INLINE: the goals over the joint values are inlined (the one hot site: secondary fitness of every child in the pre-selection)
template <bool JS_INLINE = false, class XA = XV>
BIOIK_DEV double goal_eval(ProbPtr pb, int type, int var_op, int var_seed, const double* P, const F7& fb, const XA& x, const QueryCtx& qc) {
    const int n_ops = pb->n_ops;
    switch (type) {
        case G_POSITION:  // goal_types.h:96
            return dist2(fb.p, v3(P[0], P[1], P[2]));
        case G_ORIENTATION: {  // :115-124
            const Q4 d = Q4{P[0] - fb.q.x, P[1] - fb.q.y, P[2] - fb.q.z, P[3] - fb.q.w};
            const Q4 a = Q4{P[0] + fb.q.x, P[1] + fb.q.y, P[2] + fb.q.z, P[3] + fb.q.w};
            return fmin(qdot(d, d), qdot(a, a));
        }
        case G_POSE:
            return pose_goal_cost(P, fb);
        case G_AVOID_JOINT_LIMITS:
        case G_CENTER_JOINTS:
        case G_REGULARIZATION:
        case G_MINIMAL_DISPLACEMENT:
        case G_JOINT_VARIABLE:
            if constexpr (!std::is_same<XA, XV>::value) {  // a computed accessor cannot cross a call: inline
                return goal_eval_joint_set_x(pb, type, var_op, var_seed, type == G_JOINT_VARIABLE ? P[0] : 0.0, x, (const lds_f64*)qc.seed);
            } else {
                if (JS_INLINE)
                    return goal_eval_joint_set_inl(pb, type, var_op, var_seed, type == G_JOINT_VARIABLE ? P[0] : 0.0, (const lds_f64*)x.p, x.s, (const lds_f64*)qc.seed);
                return goal_eval_joint_set(pb, type, var_op, var_seed, type == G_JOINT_VARIABLE ? P[0] : 0.0, (const lds_f64*)x.p, x.s, (const lds_f64*)qc.seed);
            }
        default:  // the remaining link goals
            return goal_eval_link_rare(type, (const lds_f64*)P, fb);
    }
    return 0.0;
}

// Σ weight² · e over the primary link goals of one tip (problem.cpp:244-257, grouped by tip)
// (sum: the fitness so far.  The reference adds goal after goal to ONE running sum (problem.cpp:244-257); a tip's goals continue the caller's sum instead of
// forming one of their own, so that (s + a) + b is what is computed, not s + (a + b) -- the same for every evaluation path of the device, and the reference's
// bits wherever the goals are listed in the order the walk completes their tips, gene-only goals behind them)
template <class XA>
BIOIK_DEV double tip_goals(ProbPtr pb, int t, const F7& f, const XA& x, const QueryCtx& qc, double sum) {
#if !defined(BIOIK_NO_POSE_ONLY)
    // the usual tip: one PoseGoal (DevTip::pose_off): the sum below without the goal table's dependent scalar loads
    if (pb->tips[t].pose_off >= 0) return sum + pose_goal_cost(qc.par + pb->tips[t].pose_off, f) * pb->tips[t].pose_weight_sq;
#endif
    const int g0 = pb->tips[t].goal_first, g1 = g0 + pb->tips[t].goal_count;
    for (int g = g0; g < g1; g++)
        sum += goal_eval<false, XA>(pb, pb->primary[g].type, pb->primary[g].var_op, pb->primary[g].var_seed, qc.par + pb->primary[g].param_off, f, x, qc) *
               pb->primary[g].weight_sq;
    return sum;
}
// primary goals that read no link
template <class XA>
BIOIK_DEV double nonlink_primary(ProbPtr pb, const XA& x, const QueryCtx& qc, double sum) {
    const F7 zero = F7{{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    for (int g = pb->n_link_primary; g < pb->n_primary; g++)
        sum += goal_eval<false, XA>(pb, pb->primary[g].type, pb->primary[g].var_op, pb->primary[g].var_seed, qc.par + pb->primary[g].param_off, zero, x, qc) *
               pb->primary[g].weight_sq;
    return sum;
}
// BalanceGoal (goal_types.cpp:257-272): the centre of mass of the robot = sum over the links with mass of (frame o centre) * share; the
// walk feeds every such tip into `acc` as its frame completes, and the goals finish on the sum.  Only the general kernel flavour carries
// it (a problem with a BalanceGoal is never handed to the lean one).
template <class PB>
BIOIK_DEV void balance_tip(PB pb, int t, const F7& f, V3& acc) {
    if constexpr (pb_flavour<PB>::general) {
        const double w = pb->tips[t].bal_w;
        if (w != 0.0) {
            BIOIK_FP_STRICT
            V3 c = qrot(f.q, v3(pb->tips[t].bal_c[0], pb->tips[t].bal_c[1], pb->tips[t].bal_c[2]));
            c = c + f.p;
            acc = acc + c * w;
        }
    }
}
template <class PB>
BIOIK_DEV double balance_goal_cost(PB pb, int g, const V3& acc, const QueryCtx& qc) {  // one BalanceGoal, weighted
    const double* P = qc.par + pb->balance[g].param_off;
    V3 center = acc - v3(P[0], P[1], P[2]);
    const V3 axis = v3(P[3], P[4], P[5]);
    center = center - axis * dot3(axis, center);
    return len2(center) * pb->balance[g].weight_sq;
}
template <class PB>
BIOIK_DEV double balance_cost(PB pb, const V3& acc, const QueryCtx& qc) {
    double sum = 0.0;
    if constexpr (pb_flavour<PB>::general)
        for (int g = 0; g < pb->n_balance; g++) sum += balance_goal_cost(pb, g, acc, qc);
    return sum;
}
// secondary goals see genes only; link goals marked secondary read null frames (ik_base.h:163)
template <bool JS_INLINE = false, class XA = XV>
BIOIK_DEV double secondary_fitness(ProbPtr pb, const XA& x, const QueryCtx& qc) {
    double sum = 0.0;
    const F7 zero = F7{{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    for (int g = 0; g < pb->n_secondary; g++)
        sum += goal_eval<JS_INLINE, XA>(pb, pb->secondary[g].type, pb->secondary[g].var_op, pb->secondary[g].var_seed, qc.par + pb->secondary[g].param_off, zero, x, qc) *
               pb->secondary[g].weight_sq;
    return sum;
}
// the same for N individuals at once (see goal_eval_joint_set_xn); per individual: the goals in their order, each weighted as above
// AvoidJointLimitsGoal (goal_types.h:387-401) costs a variable nothing while it stays in the middle half of its range: |x - mid| * 2 <= span / 2.  A child's
// gene is parent_gene + gauss * rate * span + parent_gradient * factor with |gauss| <= 4.41 (rng_gauss32: 3.451 since round 6, 4.41 = the bound of the lattice Gaussian of rounds 1 - 5), rate <= 2^-8
// (ChildX: 2^(15 - 23)) and factor <= 2, clipped towards the inside afterwards: whenever the parent's gene is further inside the free zone than
// 0.0173 span + 2 |parent_gradient| (+ a margin far above any rounding), EVERY child of the generation has that gene in the free zone.  True for about half
// the genes of a typical elite -- and for those the pre-selection need not generate the children's values at all.
BIOIK_DEV bool avoid_limits_surely_free(double parent_gene, double pg_even, double pg_odd, double vmin, double vmax, double span) {
    const double mid = (vmin + vmax) * 0.5;
    const double reach = 0.0173 * span + 2.0 * fmax(fabs(pg_even), fabs(pg_odd));
    return fabs(parent_gene - mid) + reach < 0.25 * span * (1.0 - 1e-9);
}
template <int N, class XA>
BIOIK_DEV void secondary_fitness_n(ProbPtr pb, const XA (&x)[N], const QueryCtx& qc, double (&out)[N], uint64_t inside_mask = 0ull) {
    const F7 zero = F7{{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = 0.0;
    for (int g = 0; g < pb->n_secondary; g++) {
        const int type = pb->secondary[g].type;
        const double w = pb->secondary[g].weight_sq;
        double e[N];
        if (type == G_AVOID_JOINT_LIMITS || type == G_CENTER_JOINTS || type == G_REGULARIZATION || type == G_MINIMAL_DISPLACEMENT || type == G_JOINT_VARIABLE) {
            const double* P = qc.par + pb->secondary[g].param_off;
            goal_eval_joint_set_xn<N>(pb, type, pb->secondary[g].var_op, pb->secondary[g].var_seed, type == G_JOINT_VARIABLE ? P[0] : 0.0, x, (const lds_f64*)qc.seed, e, inside_mask);
        } else {
#pragma unroll
            for (int j = 0; j < N; j++)
                e[j] = goal_eval<true, XA>(pb, type, pb->secondary[g].var_op, pb->secondary[g].var_seed, qc.par + pb->secondary[g].param_off, zero, x[j], qc);
        }
#pragma unroll
        for (int j = 0; j < N; j++) out[j] += e[j] * w;
    }
}

// ---------------------------------------------------------------------------------------------------------
// exact FK: every lane walks the joint program with the values of ITS individual (forward_kinematics.h:331-354)
//   slots      LDS, [slot][7][nthreads]: parked branch frames of this lane
//   frames_out LDS or null, [op][7]: the lane that passes it publishes every joint's global frame
//              (the per-joint frame chain the analytic Jacobian reads)
//   tip_fn(t, frame): called with device tip index t as soon as the tip's frame is complete
// The loop trip count and every constant are wave-uniform: control flow is scalar, constants arrive in SGPRs.
// ---------------------------------------------------------------------------------------------------------
// frame.h:189-209
BIOIK_DEV F7 f7_invert(const F7& a) {
    const Q4 qi = qinv(a.q);
    return F7{qrot(qi, v3(-a.p.x, -a.p.y, -a.p.z)), qi};
}
// Floating / planar joints (RobotJointEvaluator::getJointFrame, forward_kinematics.h:120-135): the joint's 7 / 3 values travel
// as an F7 (floating: translation, quaternion; planar: x, y, theta in p), and the arithmetic -- normalisation of the quaternion
// (sqrt, four divisions) or sincos, two frame concatenations -- is ONE out-of-line value function: these joints are rare, and
// every inlined copy would weigh on the instruction cache of the robots that have none.
BIOIK_DEV F7 multi_joint_values(int type, const XV& x, int val_first) {
    F7 v = F7{{x(val_first), x(val_first + 1), x(val_first + 2)}, {0.0, 0.0, 0.0, 0.0}};
    if (type == BIOIK_OP_FLOATING) v.q = Q4{x(val_first + 3), x(val_first + 4), x(val_first + 5), x(val_first + 6)};
    return v;
}
BIOIK_DEV void multi_joint_bump(F7& v, int i, double step) {  // variable i += step (forward difference of the Jacobian, :695-726)
    // (selects, not a pointer into the frame: an address-taken F7 lives in scratch memory)
    v.p.x = i == 0 ? v.p.x + step : v.p.x, v.p.y = i == 1 ? v.p.y + step : v.p.y, v.p.z = i == 2 ? v.p.z + step : v.p.z;
    v.q.x = i == 3 ? v.q.x + step : v.q.x, v.q.y = i == 4 ? v.q.y + step : v.q.y, v.q.z = i == 5 ? v.q.z + step : v.q.z, v.q.w = i >= 6 ? v.q.w + step : v.q.w;
}
BIOIK_DEV F7 multi_joint_frame(int type, const F7& v) {
    if (type == BIOIK_OP_FLOATING) {
        const double inv = 1.0 / sqrt(qdot(v.q, v.q));  // tf2 Quaternion::normalized: q * (1 / length)
        return F7{v.p, {v.q.x * inv, v.q.y * inv, v.q.z * inv, v.q.w * inv}};
    }
    double sn, cs;
    p_sincos(v.p.z * 0.5, &sn, &cs);  // planar: Translation(x, y, 0) * AngleAxis(theta, Z)
    return F7{{v.p.x, v.p.y, 0.0}, {0.0, 0.0, sn, cs}};
}
// the joint frame J(values) of a floating / planar joint (one out-of-line copy: a square root and a division, or a sincos)
BIOIK_CALL F7 multi_joint_local(int type, F7 values) { return multi_joint_frame(type, values); }
// In front of a chain walk: the joint frames of the floating / planar joints of individual x go to this lane's slots (ops[k].multi_slot); the op
// itself then applies its constant frame and fetches the joint frame -- nothing of the joint-frame arithmetic is inside the joint loop.
BIOIK_DEV void multi_joint_prologue(ProbPtr pb, const XV& x, double* slots_of_child) {
    const int mo = pb->multi_op;
    if (mo < 0) return;
    const int tid = p_tid(), nth = p_nthreads();
    const int n_chain = pb->n_chain_ops;
    for (int k = mo; k < n_chain; k++) {
        const int type = pb->ops[k].type;
        if (type < BIOIK_OP_FLOATING) continue;
        const F7 f = multi_joint_local(type, multi_joint_values(type, x, pb->ops[k].val_first));
        double* sl = slots_of_child + (size_t)pb->ops[k].multi_slot * 7 * nth + tid;
        sl[0] = f.p.x, sl[(size_t)nth] = f.p.y, sl[(size_t)2 * nth] = f.p.z;
        sl[(size_t)3 * nth] = f.q.x, sl[(size_t)4 * nth] = f.q.y, sl[(size_t)5 * nth] = f.q.z, sl[(size_t)6 * nth] = f.q.w;
    }
}
// ... and inside the walk: F = (F_src o C) o J with J from the slot (the caller has applied C)
BIOIK_DEV F7 multi_joint_fetch(const F7& f, const double* slots_of_child, int multi_slot, int nth, int tid) {
    const double* s = slots_of_child + (size_t)multi_slot * 7 * nth + tid;
    return f7_concat(f, F7{{s[0], s[(size_t)nth], s[(size_t)2 * nth]}, {s[(size_t)3 * nth], s[(size_t)4 * nth], s[(size_t)5 * nth], s[(size_t)6 * nth]}});
}

// accessors whose entry is COMPUTED from numbers of the op (ChildX, ChildT: `at`) say so
template <class XA, class = void>
struct accessor_takes_op_numbers : std::false_type {};
template <class XA>
struct accessor_takes_op_numbers<XA, std::enable_if_t<XA::takes_op_numbers>> : std::true_type {};
// value of the joint of op k in the individual x: its own entry, or for a mimic joint factor * (entry of the joint it
// follows) + offset (RobotFK_Fast_Base::updateMimic, forward_kinematics.h:230-246; a plain multiply and add)
template <class XA>
BIOIK_DEV double joint_value(const XA& x, int k, int mimic_src, double mimic_factor, double mimic_offset) {
    BIOIK_FP_STRICT
    double v = x(k);
    if (mimic_src >= 0) v = x(mimic_src) * mimic_factor + mimic_offset;
    return v;
}

//   prefix     LDS or null, [7]: the frame behind ops[0..n_prefix), which is the same for every individual of the query; the
//              walk then starts at op n_prefix (the kernels that own a query compute it once, fk_prefix)
//   COOP       lanes that walk the SAME individual together (0: every lane its own).  The solver evaluates single individuals --
//              an elite for the species ranking, the linearisation point of the memetic phase, the solution's success test -- on a
//              whole wavefront (64), or on one half of it per species (32), and in the plain walk every one of those lanes repeats
//              every joint's trigonometry.  With COOP lane k of the group takes the value of op k and its sincos ONCE, all joints at a
//              time, and the chain loop fetches the three numbers of its joint from that lane (v_readlane / ds_bpermute): the
//              composition -- the part that has to be sequential -- is what is left in the loop, about a third of a joint's
//              instructions, and the frames are the same bits (the same sincos of the same value, composed in the same order).
//              Needs n_chain_ops <= COOP (else the plain walk runs).
template <int COOP = 0, class PB, class XA, class TipFn>
BIOIK_DEV void fk_walk(PB pb, const XA& x, double* slots, double* frames_out, TipFn&& tip_fn, const double* prefix = nullptr) {
    const int nth = p_nthreads();  // (the lane's own number: p_tid_fresh() where a parked frame is addressed -- rare, and nothing is carried through the walk for it)
    const int n_chain = pb->n_chain_ops;
    F7 f = f7_identity();
    for (int t = 0; t < pb->n_root_tips; t++) {
        if (pb->tips[t].has_e) {
            double e[7];
            for (int c = 0; c < 7; c++) e[c] = pb->tips[t].e[c];
            tip_fn(t, f7_load(e));
        } else {
            tip_fn(t, f);
        }
    }
    int k_begin = 0;
    if (prefix) {
        k_begin = pb->n_prefix;
        if (k_begin > 0) f = f7_load(prefix);
    }
    if constexpr (pb_flavour<PB>::general && std::is_same<XA, XV>::value) multi_joint_prologue(pb, x, slots);
    // COOP: this lane's joint, its value and half-angle trigonometry (computed for prismatic joints too and discarded: no branch)
    const bool coop = COOP > 0 && n_chain <= COOP;
    const int lane = COOP > 0 ? p_lane_fresh() : 0;  // (= tid & 63, computed here: the walk itself keeps no lane number alive)
    const int coop_base = COOP > 0 ? (lane & ~(COOP - 1)) : 0;  // first lane of this lane's group inside its wavefront
    double my_xv = 0.0, my_sn = 0.0, my_cs = 1.0;
    if (coop) {
        const int mine = lane - coop_base, kk = mine < n_chain ? mine : n_chain - 1;
        my_xv = joint_value(x, kk, pb->ops[kk].mimic_src, pb->ops[kk].mimic_factor, pb->ops[kk].mimic_offset);
        p_sincos(my_xv * 0.5, &my_sn, &my_cs);
    }
    for (int k = k_begin; k < n_chain; k++) {
        double xv, sn, cs;
        if (coop) {
            if (COOP >= 64) xv = p_read_lane(my_xv, k), sn = p_read_lane(my_sn, k), cs = p_read_lane(my_cs, k);
            else xv = p_shfl(my_xv, coop_base + k), sn = p_shfl(my_sn, coop_base + k), cs = p_shfl(my_cs, coop_base + k);
        } else {
            xv = joint_value(x, k, pb->ops[k].mimic_src, pb->ops[k].mimic_factor, pb->ops[k].mimic_offset);
            p_sincos(xv * 0.5, &sn, &cs);
        }
        const int type = pb->ops[k].type, src = pb->ops[k].src, ls = pb->ops[k].load_slot, ss = pb->ops[k].save_slot;
        const int t0 = pb->ops[k].tip_first, t1 = t0 + pb->ops[k].tip_count;
        const double ca0 = pb->ops[k].ca[0], ca1 = pb->ops[k].ca[1], ca2 = pb->ops[k].ca[2], ca3 = pb->ops[k].ca[3];
        const double cb0 = pb->ops[k].cb[0], cb1 = pb->ops[k].cb[1], cb2 = pb->ops[k].cb[2], cb3 = pb->ops[k].cb[3];
        const double cp0 = pb->ops[k].cpos[0], cp1 = pb->ops[k].cpos[1], cp2 = pb->ops[k].cpos[2];
        if (ls >= 0) {
            const double* s = slots + (size_t)ls * 7 * nth + p_tid_fresh();
            f = F7{{s[0], s[(size_t)nth], s[(size_t)2 * nth]}, {s[(size_t)3 * nth], s[(size_t)4 * nth], s[(size_t)5 * nth], s[(size_t)6 * nth]}};
        } else if (k > 0 && src < 0) {
            f = f7_identity();
        }
        if (type == BIOIK_OP_REVOLUTE) {  // (wavefront-uniform: a scalar branch)
            F7 fj[1] = {f};
            const double s1[1] = {sn}, c1[1] = {cs};
            revolute_apply<1>(fj, s1, c1, RevConst{cp0, cp1, cp2, ca0, ca1, ca2, ca3, cb0, cb1, cb2, cb3, pb->ops[k].pos_kind, pb->ops[k].rot_kind});
            f = fj[0];
        } else {
            const V3 lp = v3(BK_FMA(xv, cb0, cp0), BK_FMA(xv, cb1, cp1), BK_FMA(xv, cb2, cp2));
            f.p = f.p + qrot(f.q, lp);
            f.q = qmul(f.q, Q4{ca0, ca1, ca2, ca3});
            if constexpr (pb_flavour<PB>::general)
                if (type >= BIOIK_OP_FLOATING) f = multi_joint_fetch(f, slots, pb->ops[k].multi_slot, nth, p_tid_fresh());
        }
        if (ss >= 0) {
            double* sl = slots + (size_t)ss * 7 * nth + p_tid_fresh();
            sl[0] = f.p.x;
            sl[(size_t)nth] = f.p.y;
            sl[(size_t)2 * nth] = f.p.z;
            sl[(size_t)3 * nth] = f.q.x;
            sl[(size_t)4 * nth] = f.q.y;
            sl[(size_t)5 * nth] = f.q.z;
            sl[(size_t)6 * nth] = f.q.w;
        }
        if (frames_out) f7_store(frames_out + k * 7, f);  // only the publishing lane passes a non-null pointer
        for (int t = t0; t < t1; t++) {
            if (pb->tips[t].has_e) {
                double e[7];
                for (int c2 = 0; c2 < 7; c2++) e[c2] = pb->tips[t].e[c2];
                tip_fn(t, f7_concat(f, f7_load(e)));
            } else {
                tip_fn(t, f);
            }
        }
    }
}

// primary fitness on the exact-FK phenotype (ik_base.h:203-207 semantics, per individual)
// the frame behind the leading run of non-gene joints (DevProblem::n_prefix) at the values x: exactly the frame fk_walk
// holds after those joints
BIOIK_DEV F7 fk_prefix(ProbPtr pb, const XV& x) {
    F7 f = f7_identity();
    const int n = pb->n_prefix;
    for (int k = 0; k < n; k++) {
        double sn, cs;
        const double xv = x(k);
        p_sincos(xv * 0.5, &sn, &cs);
        const bool rev = pb->ops[k].type == BIOIK_OP_REVOLUTE;
        const double s = rev ? sn : 0.0, c = rev ? cs : 1.0, xp = rev ? 0.0 : xv;
        const Q4 lq = Q4{BK_FMA(c, pb->ops[k].ca[0], s * pb->ops[k].cb[0]), BK_FMA(c, pb->ops[k].ca[1], s * pb->ops[k].cb[1]),
                         BK_FMA(c, pb->ops[k].ca[2], s * pb->ops[k].cb[2]), BK_FMA(c, pb->ops[k].ca[3], s * pb->ops[k].cb[3])};
        const V3 lp = v3(BK_FMA(xp, pb->ops[k].cb[0], pb->ops[k].cpos[0]), BK_FMA(xp, pb->ops[k].cb[1], pb->ops[k].cpos[1]),
                         BK_FMA(xp, pb->ops[k].cb[2], pb->ops[k].cpos[2]));
        f.p = f.p + qrot(f.q, lp);
        f.q = qmul(f.q, lq);
    }
    return f;
}

template <class PB, class XA>
BIOIK_DEV double eval_exact_primary(PB pb, const XA& x, const QueryCtx& qc, double* slots, const double* prefix = nullptr) {
    double sum = 0.0;
    V3 bal = v3(0.0, 0.0, 0.0);
    fk_walk(pb, x, slots, nullptr, [&](int t, const F7& f) {
        sum = tip_goals(pb, t, f, x, qc, sum);
        balance_tip(pb, t, f, bal);
    }, prefix);
    sum = nonlink_primary(pb, x, qc, sum);
    sum += balance_cost(pb, bal, qc);
    return sum;
}

// N individuals per lane at once (the children a lane owns in one generation): the same walk with N independent
// dependency chains, so that the scalar loads of a joint's constants, the LDS reads of the gene values and the latency of the
// polynomial chains are paid once per joint instead of once per joint and child.  Arithmetic per individual is identical
// to fk_walk.  Parked branch frames: child j uses the slot set at slots + j * slot_set_stride.
// SERIAL (DevProblem::serial_chain, the caller's promise): no op fetches or parks a branch frame, restarts at the root or mimics another joint -- the
// loop body then has ONE definition of the running frames, and the compiler keeps them in the same registers from joint to joint (with the branch
// paths in the body it copies all 14 N numbers between two sets of registers in every trip)
template <int N, bool SERIAL = false, class PB, class XA, class TipFn>
BIOIK_DEV void fk_walk_n(PB pb, const XA (&x)[N], double* slots, int slot_set_stride, TipFn&& tip_fn, const double* prefix = nullptr) {
    const int nth = p_nthreads();
    const int n_chain = pb->n_chain_ops;
    F7 f[N];
#pragma unroll
    for (int j = 0; j < N; j++) f[j] = f7_identity();
    for (int t = 0; t < pb->n_root_tips; t++) {
        F7 o[N];
        if (pb->tips[t].has_e) {
            double e[7];
            for (int c = 0; c < 7; c++) e[c] = pb->tips[t].e[c];
#pragma unroll
            for (int j = 0; j < N; j++) o[j] = f7_load(e);
        } else {
#pragma unroll
            for (int j = 0; j < N; j++) o[j] = f[j];
        }
        tip_fn(t, o);
    }
    int k_begin = 0;
    if (prefix) {
        k_begin = pb->n_prefix;
        if (k_begin > 0) {
            const F7 f0 = f7_load(prefix);
#pragma unroll
            for (int j = 0; j < N; j++) f[j] = f0;
        }
    }
    if constexpr (pb_flavour<PB>::general && std::is_same<XA, XV>::value)
        for (int j = 0; j < N; j++) multi_joint_prologue(pb, x[j], slots + (size_t)j * slot_set_stride);
    for (int k = k_begin; k < n_chain; k++) {
        // every scalar of the joint is requested here, in one burst of scalar loads that is waited for once (reading
        // them where they are used costs one exposed scalar-cache round trip per branch of the loop body)
        const int type = pb->ops[k].type, src = SERIAL ? k - 1 : pb->ops[k].src, ls = SERIAL ? -1 : pb->ops[k].load_slot, ss = SERIAL ? -1 : pb->ops[k].save_slot;
        const int pk = pb->ops[k].pos_kind, rk = pb->ops[k].rot_kind;
        const int t0 = pb->ops[k].tip_first, t1 = t0 + pb->ops[k].tip_count;
        const int msrc = SERIAL ? -1 : pb->ops[k].mimic_src;
        const double mf = SERIAL ? 1.0 : pb->ops[k].mimic_factor, mo = SERIAL ? 0.0 : pb->ops[k].mimic_offset;
        const double ca0 = pb->ops[k].ca[0], ca1 = pb->ops[k].ca[1], ca2 = pb->ops[k].ca[2], ca3 = pb->ops[k].ca[3];
        const double cb0 = pb->ops[k].cb[0], cb1 = pb->ops[k].cb[1], cb2 = pb->ops[k].cb[2], cb3 = pb->ops[k].cb[3];
        const double cp0 = pb->ops[k].cpos[0], cp1 = pb->ops[k].cpos[1], cp2 = pb->ops[k].cpos[2];
        double xv[N];
        if constexpr (SERIAL && accessor_takes_op_numbers<XA>::value) {
            // (children computed where they are read: what their accessor reads of the op comes with the burst above)
            const int gene = pb->ops[k].gene;
            const double span = pb->ops[k].span, cmin = pb->ops[k].clip_min, cmax = pb->ops[k].clip_max;
#pragma unroll
            for (int j = 0; j < N; j++) xv[j] = x[j].at(k, gene, span, cmin, cmax);
        } else {
#pragma unroll
            for (int j = 0; j < N; j++) xv[j] = joint_value(x[j], k, msrc, mf, mo);
        }
        if (ls >= 0) {
            const int tid = p_tid_fresh();
#pragma unroll
            for (int j = 0; j < N; j++) {
                const double* s = slots + (size_t)j * slot_set_stride + (size_t)ls * 7 * nth + tid;
                f[j] = F7{{s[0], s[(size_t)nth], s[(size_t)2 * nth]}, {s[(size_t)3 * nth], s[(size_t)4 * nth], s[(size_t)5 * nth], s[(size_t)6 * nth]}};
            }
        } else if (k > 0 && src < 0) {
#pragma unroll
            for (int j = 0; j < N; j++) f[j] = f7_identity();
        }
        // the joint's local frame: a revolute joint turns about its axis behind the constant frame (position = the constant's), every
        // other op slides along cb (zero for the ops that only fetch a parked frame).  The op type is wavefront-uniform: a scalar
        // branch, not a select per component; the half-angle trigonometry only where it is used,
        // and the frame applied inside each branch so that the constants stay scalar operands
        if (type == BIOIK_OP_REVOLUTE) {
            double sn[N], cs[N];
#pragma unroll
            for (int j = 0; j < N; j++) p_sincos(xv[j] * 0.5, &sn[j], &cs[j]);
            revolute_apply<N>(f, sn, cs, RevConst{cp0, cp1, cp2, ca0, ca1, ca2, ca3, cb0, cb1, cb2, cb3, pk, rk});
        } else {
#pragma unroll
            for (int j = 0; j < N; j++) {
                const V3 lp = v3(BK_FMA(xv[j], cb0, cp0), BK_FMA(xv[j], cb1, cp1), BK_FMA(xv[j], cb2, cp2));
                f[j].p = f[j].p + qrot(f[j].q, lp);
                f[j].q = qmul(f[j].q, Q4{ca0, ca1, ca2, ca3});
            }
            if constexpr (pb_flavour<PB>::general)
                if (type >= BIOIK_OP_FLOATING) {
                    const int tid = p_tid_fresh();
#pragma unroll
                    for (int j = 0; j < N; j++) f[j] = multi_joint_fetch(f[j], slots + (size_t)j * slot_set_stride, pb->ops[k].multi_slot, nth, tid);
                }
        }
        if (ss >= 0) {
            const int tid = p_tid_fresh();
#pragma unroll
            for (int j = 0; j < N; j++) {
                double* sl = slots + (size_t)j * slot_set_stride + (size_t)ss * 7 * nth + tid;
                sl[0] = f[j].p.x;
                sl[(size_t)nth] = f[j].p.y;
                sl[(size_t)2 * nth] = f[j].p.z;
                sl[(size_t)3 * nth] = f[j].q.x;
                sl[(size_t)4 * nth] = f[j].q.y;
                sl[(size_t)5 * nth] = f[j].q.z;
                sl[(size_t)6 * nth] = f[j].q.w;
            }
        }
        for (int t = t0; t < t1; t++) {
            F7 o[N];
            if (pb->tips[t].has_e) {
                double e[7];
                for (int c2 = 0; c2 < 7; c2++) e[c2] = pb->tips[t].e[c2];
#pragma unroll
                for (int j = 0; j < N; j++) o[j] = f7_concat(f[j], f7_load(e));
            } else {
#pragma unroll
                for (int j = 0; j < N; j++) o[j] = f[j];
            }
            tip_fn(t, o);
        }
    }
}
// LINKS_ONLY: the walk and the goals that read a tip; the caller adds nonlink_primary and balance_cost itself (the dense kernel: it re-derives
// the children's accessors behind the walk instead of carrying them through it)
template <int N, bool LINKS_ONLY = false, bool SERIAL = false, class PB, class XA>
BIOIK_DEV void eval_exact_primary_n(PB pb, const XA (&x)[N], const QueryCtx& qc, double* slots, int slot_set_stride, double (&out)[N],
                                    const double* prefix = nullptr) {
    V3 bal[N];
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = 0.0, bal[j] = v3(0.0, 0.0, 0.0);
    fk_walk_n<N, SERIAL>(pb, x, slots, slot_set_stride, [&](int t, const F7 (&f)[N]) {
#pragma unroll
        for (int j = 0; j < N; j++) balance_tip(pb, t, f[j], bal[j]);
        // the goals of the tip, each evaluated for the N individuals (per individual: the summation order of tip_goals).  (No one-PoseGoal
        // form here, unlike tip_goals: this loop is bound by issue slots, not by the latency of the goal table, and the second path costs
        // the generation loop four registers -- three spilled values in the computed-children kernel, ten more under its 128-register budget.)
        const int g0 = pb->tips[t].goal_first, g1 = g0 + pb->tips[t].goal_count;
        for (int g = g0; g < g1; g++) {
            const int type = pb->primary[g].type, var_op = pb->primary[g].var_op, var_seed = pb->primary[g].var_seed;
            // (p_fresh: the goal's numbers are read HERE, when a tip's frame is complete -- as a loop invariant the compiler reads them in front of the
            // joint loop and carries up to sixteen registers of them through every joint of the walk)
            const int po = p_fresh(pb->primary[g].param_off);
            const double w = pb->primary[g].weight_sq;
            const double* P = qc.par + po;
#pragma unroll
            for (int j = 0; j < N; j++) out[j] += goal_eval<false, XA>(pb, type, var_op, var_seed, P, f[j], x[j], qc) * w;
        }
    }, prefix);
    if constexpr (!LINKS_ONLY) {
#pragma unroll
        for (int j = 0; j < N; j++) out[j] = nonlink_primary(pb, x[j], qc, out[j]), out[j] += balance_cost(pb, bal[j], qc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// linearised phenotype model (RobotFK_Mutator): tables in LDS
//   tipbase [T][7] tip frames at the base configuration, delta [T][n_ops][7] per-(tip,op) first-order frames,
//   base [n_ops] op values at the base configuration
// ---------------------------------------------------------------------------------------------------------
struct LinModel {
    const double* tipbase;
    const double* delta;
    const double* base;
};

// forward_kinematics.h:1186-1231 (no renormalisation of the quaternion).  The sum runs over the GENES in gene order, as the
// reference's does (approx_map holds gene indices in ascending order; a gene that does not move the tip contributes d * dv with
// d = 0, an exact no-op).  Where the ops meet the genes in that same order (DevProblem::genes_follow_ops: every robot without
// floating / planar joints so far) the walk is over the ops and needs no gene -> op look-up; otherwise over op_of_gene.
// Four entries per trip: their 4 + 4 + 28 LDS operands are requested together and waited for once; padding contributes d * 0.0.
template <class PB, class XA>
BIOIK_DEV F7 linear_tip(PB pb, int t, const XA& x, const LinModel& lm) {
    const int n_ops = pb->n_ops;
    const bool by_op = pb_flavour<PB>::general ? pb->genes_follow_ops != 0 : true;
    const int cnt = by_op ? n_ops : pb->D;
    const uint64_t active = pb->active_mask;
    const double* tb = lm.tipbase + t * 7;
    double px = tb[0], py = tb[1], pz = tb[2], rx = tb[3], ry = tb[4], rz = tb[5], rw = tb[6];
    for (int g0 = 0; g0 < cnt; g0 += 4) {
        double dv[4], d[4][7];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int idx = g0 + j < cnt ? g0 + j : cnt - 1;
            const int kk = by_op ? idx : pb->op_of_gene[idx];
            const bool on = g0 + j < cnt && ((active >> kk) & 1ull);
            const double xv = x(kk), bv = lm.base[kk];
            dv[j] = on ? xv - bv : 0.0;
            const double* dp = lm.delta + ((size_t)t * n_ops + kk) * 7;
#pragma unroll
            for (int c = 0; c < 7; c++) d[j][c] = dp[c];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            px = BK_FMA(d[j][0], dv[j], px);
            py = BK_FMA(d[j][1], dv[j], py);
            pz = BK_FMA(d[j][2], dv[j], pz);
            rx = BK_FMA(d[j][3], dv[j], rx);
            ry = BK_FMA(d[j][4], dv[j], ry);
            rz = BK_FMA(d[j][5], dv[j], rz);
            rw = BK_FMA(d[j][6], dv[j], rw);
        }
    }
    return F7{{px, py, pz}, {rx, ry, rz, rw}};
}

template <class PB, class XA>
BIOIK_DEV double eval_linear_primary(PB pb, const XA& x, const QueryCtx& qc, const LinModel& lm) {
    double sum = 0.0;
    const int T = pb->T;
    V3 bal = v3(0.0, 0.0, 0.0);
    for (int t = 0; t < T; t++) {
        const F7 f = linear_tip(pb, t, x, lm);
        sum = tip_goals(pb, t, f, x, qc, sum);
        balance_tip(pb, t, f, bal);
    }
    sum = nonlink_primary(pb, x, qc, sum);
    sum += balance_cost(pb, bal, qc);
    return sum;
}

// One (tip, op) entry of the approximator tables from the published joint frames:
// tip-local Jacobian column (forward_kinematics.h:639-693) -> world delta frame (:827-852).
// tip-local Jacobian column of ONE joint (op m) for tip t: linear and angular part (forward_kinematics.h:639-693)
BIOIK_DEV void jacobian_column(ProbPtr pb, int m, const F7& lf, const F7& tf, V3& vel, V3& om) {
    V3 axis = v3(pb->ops[m].axis[0], pb->ops[m].axis[1], pb->ops[m].axis[2]);
    // tf2 Quaternion product inverse(link.rot) * tip.rot, then inverse
    Q4 a = qinv(lf.q), b = tf.q;
    Q4 q = Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
    q = qinv(q);
    V3 rot = qrot(q, axis);
    if (pb->ops[m].type == BIOIK_OP_REVOLUTE) {
        V3 d = qrot(qinv(tf.q), lf.p - tf.p);
        vel = cross3(d, rot);
        om = rot;
    } else {
        vel = rot;
        om = v3(0.0, 0.0, 0.0);
    }
}
// frame.h:240-259: twist of b seen from a (translation, rotation vector), tf2 getAngle / getAxis semantics
BIOIK_DEV void frame_twist(const F7& a, const F7& b, V3& lin, V3& ang) {
    const F7 f = f7_concat(f7_invert(a), b);
    lin = f.p;
    double ra = 2.0 * clamped_acos(f.q.w);
    if (ra > +BIOIK_PI) ra -= 2 * BIOIK_PI;
    const double s_squared = 1.0 - f.q.w * f.q.w;
    V3 axis = v3(1.0, 0.0, 0.0);
    if (!(s_squared < 10.0 * 2.220446049250313e-16)) {
        const double s = 1.0 / sqrt(s_squared);
        axis = v3(f.q.x * s, f.q.y * s, f.q.z * s);
    }
    ang = axis * ra;
}
// Jacobian column of ONE variable of a floating / planar joint by forward difference (forward_kinematics.h:695-726):
// values in, values out, one out-of-line copy.  parent_c = (frame of the parent op) o (constant frame in front of the joint).
struct Twist6 {
    double v[6];
};
BIOIK_CALL Twist6 jacobian_numeric(int type, F7 parent_c, F7 values_2, F7 link_frame_1, F7 tip_frame_1) {
    const double inv_step_size = 1.0 / 0.00001;
    const F7 link_frame_2 = f7_concat(parent_c, multi_joint_frame(type, values_2));
    const F7 tip_frame_2 = f7_concat(f7_concat(link_frame_2, f7_invert(link_frame_1)), tip_frame_1);  // change(), frame.h:203-209
    V3 lin, ang;
    frame_twist(tip_frame_1, tip_frame_2, lin, ang);
    Twist6 t;
    t.v[0] = lin.x * inv_step_size, t.v[1] = lin.y * inv_step_size, t.v[2] = lin.z * inv_step_size;
    t.v[3] = ang.x * inv_step_size, t.v[4] = ang.y * inv_step_size, t.v[5] = ang.z * inv_step_size;
    return t;
}
// (frame of the op in front of chain op jop) o (its constant frame): what a floating / planar joint's frame is applied to -- the arithmetic of the walk
// (frames: the published per-joint chain; prefix: the frame behind the leading non-gene joints, whose own frames are not published)
template <class PB>
BIOIK_DEV F7 multi_joint_parent_c(PB pb, int jop, const double* frames, const double* prefix) {
    const int src = pb->ops[jop].src;
    F7 f = f7_identity();
    if (src >= 0) f = (prefix != nullptr && src < pb->n_prefix) ? f7_load(prefix) : f7_load(frames + src * 7);
    return f7_concat(f, F7{{pb->ops[jop].cpos[0], pb->ops[jop].cpos[1], pb->ops[jop].cpos[2]}, {pb->ops[jop].ca[0], pb->ops[jop].ca[1], pb->ops[jop].ca[2], pb->ops[jop].ca[3]}});
}
//   base   LDS, [n_ops]: the op values of the configuration the frames were published for
//   prefix LDS or null: see fk_walk (the frames of ops[0..n_prefix) are then not published; their last one is *prefix)
template <class PB>
BIOIK_DEV void approximator_entry(PB pb, int t, int k, const double* frames, const double* tips, double* out7, const double* base = nullptr,
                                  const double* prefix = nullptr) {
    BIOIK_FP_STRICT
    const uint64_t dep_mask = pb->tips[t].dep_mask;
    const int n_chain = pb->n_chain_ops;
    const bool gene = pb->ops[k].gene >= 0;
    const int jop = pb_flavour<PB>::general ? pb->ops[k].joint_op : -1;
    if (jop >= 0) {  // a variable of a floating / planar joint
        if (!gene || ((dep_mask >> jop) & 1ull) == 0 || base == nullptr) {
            for (int c = 0; c < 7; c++) out7[c] = 0.0;
            return;
        }
        const int type = pb->ops[jop].type, vf = pb->ops[jop].val_first;
        const F7 cf = multi_joint_parent_c(pb, jop, frames, prefix);
        const F7 tf = f7_load(tips + t * 7);
        F7 values_2 = multi_joint_values(type, XV{base, 1}, vf);
        multi_joint_bump(values_2, k - vf, 0.00001);
        const Twist6 tw = jacobian_numeric(type, cf, values_2, f7_load(frames + jop * 7), tf);
        const V3 dp = qrot(tf.q, v3(tw.v[0], tw.v[1], tw.v[2]));
        const Q4 dq = qmul(tf.q, Q4{tw.v[3] * 0.5, tw.v[4] * 0.5, tw.v[5] * 0.5, 1.0});
        out7[0] = dp.x, out7[1] = dp.y, out7[2] = dp.z;
        out7[3] = dq.x - tf.q.x, out7[4] = dq.y - tf.q.y, out7[5] = dq.z - tf.q.z, out7[6] = dq.w - tf.q.w;
        return;
    }
    const bool own = gene && k < n_chain && ((dep_mask >> k) & 1ull) != 0;
    const uint64_t followers = gene ? pb->mimic_followers[k] & dep_mask : 0ull;  // mimic joints of this gene on the tip's chain
    if (!own && followers == 0ull) {
        for (int c = 0; c < 7; c++) out7[c] = 0.0;
        return;
    }
    F7 tf = f7_load(tips + t * 7);
    V3 vel = v3(0.0, 0.0, 0.0), om = v3(0.0, 0.0, 0.0);
    if (own) jacobian_column(pb, k, f7_load(frames + k * 7), tf, vel, om);
    // the joints that mimic this one move with it: their columns, scaled (joint_dependencies, forward_kinematics.h:623-636)
    for (uint64_t rest = followers; rest != 0ull; rest &= rest - 1ull) {
        const int m = __builtin_ctzll(rest);
        V3 v2, o2;
        jacobian_column(pb, m, f7_load(frames + m * 7), tf, v2, o2);
        const double scale = pb->ops[m].mimic_factor;
        vel = v3(vel.x + v2.x * scale, vel.y + v2.y * scale, vel.z + v2.z * scale);
        om = v3(om.x + o2.x * scale, om.y + o2.y * scale, om.z + o2.z * scale);
    }
    V3 dp = qrot(tf.q, vel);
    Q4 dq = qmul(tf.q, Q4{om.x * 0.5, om.y * 0.5, om.z * 0.5, 1.0});
    out7[0] = dp.x;
    out7[1] = dp.y;
    out7[2] = dp.z;
    out7[3] = dq.x - tf.q.x;
    out7[4] = dq.y - tf.q.y;
    out7[5] = dq.z - tf.q.z;
    out7[6] = dq.w - tf.q.w;
}

// The raw Jacobian column of gene op k for tip t (RobotFK_Jacobian::computeJacobian, forward_kinematics.h:600-730): tip-local linear and
// angular velocity, the six numbers the `jac` solver stacks into its least-squares system (ik_gradient.cpp:97-113).  Same frames, same
// arithmetic as approximator_entry up to the point where that one turns the column into a world-frame delta.
template <class PB>
BIOIK_DEV void jacobian_entry6(PB pb, int t, int k, const double* frames, const double* tips, double* out6, const double* base, const double* prefix = nullptr) {
    BIOIK_FP_STRICT
    const uint64_t dep_mask = pb->tips[t].dep_mask;
    const int n_chain = pb->n_chain_ops;
    const bool gene = pb->ops[k].gene >= 0;
    const int jop = pb_flavour<PB>::general ? pb->ops[k].joint_op : -1;
    for (int c = 0; c < 6; c++) out6[c] = 0.0;
    if (jop >= 0) {
        if (!gene || ((dep_mask >> jop) & 1ull) == 0) return;
        const int type = pb->ops[jop].type, vf = pb->ops[jop].val_first;
        const F7 cf = multi_joint_parent_c(pb, jop, frames, prefix);
        F7 values_2 = multi_joint_values(type, XV{base, 1}, vf);
        multi_joint_bump(values_2, k - vf, 0.00001);
        const Twist6 tw = jacobian_numeric(type, cf, values_2, f7_load(frames + jop * 7), f7_load(tips + t * 7));
        for (int c = 0; c < 6; c++) out6[c] = tw.v[c];
        return;
    }
    const bool own = gene && k < n_chain && ((dep_mask >> k) & 1ull) != 0;
    const uint64_t followers = gene ? pb->mimic_followers[k] & dep_mask : 0ull;
    if (!own && followers == 0ull) return;
    const F7 tf = f7_load(tips + t * 7);
    V3 vel = v3(0.0, 0.0, 0.0), om = v3(0.0, 0.0, 0.0);
    if (own) jacobian_column(pb, k, f7_load(frames + k * 7), tf, vel, om);
    for (uint64_t rest = followers; rest != 0ull; rest &= rest - 1ull) {
        const int m = __builtin_ctzll(rest);
        V3 v2, o2;
        jacobian_column(pb, m, f7_load(frames + m * 7), tf, v2, o2);
        const double scale = pb->ops[m].mimic_factor;
        vel = v3(vel.x + v2.x * scale, vel.y + v2.y * scale, vel.z + v2.z * scale);
        om = v3(om.x + o2.x * scale, om.y + o2.y * scale, om.z + o2.z * scale);
    }
    out6[0] = vel.x, out6[1] = vel.y, out6[2] = vel.z, out6[3] = om.x, out6[4] = om.y, out6[5] = om.z;
}

// ---------------------------------------------------------------------------------------------------------
// reproduction of ONE child (ik_evolution_2.cpp:263-300) with the counter RNG; bit-exact against the oracle.
//   p0g / p0d / p1d: LDS, op-indexed genes of parent 0 and momentum ("gradients") of parents 0 and 1
//   xo / go (stride xs / gs): where the child's genes / momentum go; go may be null
// ---------------------------------------------------------------------------------------------------------
// ik_evolution_2.cpp:320-324: the orientation genes of floating joints are pulled back towards unit length after the
// mutation (normalizeFast, frame.h:231-238: one Newton step); the momentum keeps the un-normalised difference
BIOIK_DEV void renormalize_quaternion_genes(ProbPtr pb, double* xo, int xs) {
    const int nq = pb->n_quat;
    for (int i = 0; i < nq; i++) {
        const int k = pb->quat_op[i];
        const Q4 q = Q4{xo[(size_t)k * xs], xo[(size_t)(k + 1) * xs], xo[(size_t)(k + 2) * xs], xo[(size_t)(k + 3) * xs]};
        const double f = (3.0 - qdot(q, q)) * 0.5;
        xo[(size_t)k * xs] = q.x * f, xo[(size_t)(k + 1) * xs] = q.y * f, xo[(size_t)(k + 2) * xs] = q.z * f, xo[(size_t)(k + 3) * xs] = q.w * f;
    }
}

// M independent Philox2x32-10 streams advanced in lock step: the ten rounds of one stream are a dependent chain of
// 32x32->64 multiplies, so interleaving several of them gives the in-order wavefront something to issue every cycle
template <int M>
BIOIK_DEV void philox2x32_10_xm(uint32_t key, const uint32_t (&c0in)[M], uint32_t c1in, uint32_t (&o0)[M], uint32_t (&o1)[M]) {
    uint32_t c0[M], c1[M];
#pragma unroll
    for (int j = 0; j < M; j++) c0[j] = c0in[j], c1[j] = c1in;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        if (r > 0) key += 0x9E3779B9u;
#pragma unroll
        for (int j = 0; j < M; j++) {
            uint64_t p = (uint64_t)0xD256D193u * (uint64_t)c0[j];
            uint32_t hi = (uint32_t)(p >> 32), lo = (uint32_t)p;
            c0[j] = hi ^ key ^ c1[j];
            c1[j] = lo;
        }
    }
#pragma unroll
    for (int j = 0; j < M; j++) o0[j] = c0[j], o1[j] = c1[j];
}

// Reproduction (ik_evolution_2.cpp:242-326) of N children of one lane at once (same parents, same generation): per trip
// 8 random words per child (the rate exponent + 7 genes in the first trip, 8 genes in the following ones), each one hash of
// the child's counter, the parents' genes / momentum and the joint limits loaded once per gene for all N.  Free of
// divergent branches (padding words past the last gene recompute gene D-1 and are not stored).  `go` (optional): momentum
// of child 0 (:299).
template <int N, class PB>
BIOIK_DEV void reproduce_children(PB pb, uint32_t key, uint32_t ctr1, const uint32_t (&child_index)[N], const double* p0g, const double* p0d,
                                  const double* p1d, double* const (&xo)[N], int xs, double* go = nullptr, int gs = 0) {
    BIOIK_FP_STRICT
    const int n_ops = pb->n_ops, D = pb->D;
    double fmix[N], gradient_factor[N], mutation_rate[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        fmix[i] = (child_index[i] % 2u == 0u) ? 0.2 : 0.0;
        gradient_factor[i] = (double)(child_index[i] % 3u);
        mutation_rate[i] = 0.0;
    }
    const uint32_t stream = rng_child_stream(key, ctr1);
    uint32_t base[N];
#pragma unroll
    for (int i = 0; i < N; i++) base[i] = rng_child_base(stream, child_index[i]);
    for (int w0 = 0; D > 0 && w0 <= D; w0 += 8) {  // (a problem whose variables are all fixed has no gene to draw)
        if (w0 == 0) {
#pragma unroll
            for (int i = 0; i < N; i++) mutation_rate[i] = rng_child_rate(base[i]);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (w0 + 4 * h > D) break;  // no gene in this half
            int kk[4];
            double gene[N][4], mom[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int w = 4 * h + j;  // word of this trip; its gene is w0 + w - 1
                const int g = w0 + w - 1 < 0 ? 0 : (w0 + w - 1 < D ? w0 + w - 1 : D - 1);
                const int k = pb->op_of_gene[g];
                kk[j] = k;
                const double span = pb->ops[k].span, cmin = pb->ops[k].clip_min, cmax = pb->ops[k].clip_max;
                const double parent_gene = p0g[k], d0 = p0d[k], d1 = p1d[k];
#pragma unroll
                for (int i = 0; i < N; i++) {
                    const uint32_t word = rng_child_word(base[i], (uint32_t)(w0 + w));
                    double r = rng_gauss32(word);
                    double f = mutation_rate[i] * span;
                    double gn = parent_gene;
                    gn += r * f;
                    double parent_gradient = d0 * (1.0 - fmix[i]) + d1 * fmix[i];
                    double g2 = parent_gradient * gradient_factor[i];
                    gn += g2;
                    gn = fmin(fmax(gn, cmin), cmax);
                    gene[i][j] = gn;
                    if (i == 0) mom[j] = parent_gradient * (1.0 - 0.3) + (gn - parent_gene) * 0.3;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int g = w0 + 4 * h + j - 1;
                if (g >= 0 && g < D) {
#pragma unroll
                    for (int i = 0; i < N; i++) xo[i][(size_t)kk[j] * xs] = gene[i][j];
                    if (go) go[(size_t)kk[j] * gs] = mom[j];
                }
            }
        }
    }
    if (D < n_ops)
        for (int k = 0; k < n_ops; k++)
            if (pb->ops[k].gene < 0) {
                const double v = p0g[k];  // inactive op: the seed's value, carried by every elite
#pragma unroll
                for (int i = 0; i < N; i++) xo[i][(size_t)k * xs] = v;
                if (go) go[(size_t)k * gs] = 0.0;
            }
    if constexpr (pb_flavour<PB>::general)
        for (int i = 0; i < N; i++) renormalize_quaternion_genes(pb, xo[i], xs);
}
// A child of the current generation as a FUNCTION of (parents, counter): operator()(k) computes the value of op k where it is read --
// the same operations, in the same order, as reproduce_children writes into a genotype column -- so a child needs no column in LDS.
// For problems whose columns are what limits the wavefronts per CU (many joints, large populations) the launcher picks this form: every
// gene is then hashed twice when secondary goals pre-select the children (once for the secondary fitness, once for the chain walk),
// which costs a few dozen integer instructions per gene against 8 * n_ops bytes of LDS per lane.  Lean flavour only (no quaternion genes).
template <class PB>
struct ChildX {
    PB pb;
    const double *p0g, *p0d, *p1d;  // LDS, op-indexed: genes of parent 0, momentum of parents 0 and 1
    uint32_t base;                  // the child's word 0 (rng_child_base)
    double mutation_rate, fmix, gradient_factor;
    // UNIFORM: the op index is the same in every lane of the wavefront (a loop counter of the chain walk): the clip range then comes as scalar operands
    // (p_clamp_uniform); false where lane k asks for op k (the winners' re-derivation): the same two instructions on vector operands
    template <bool UNIFORM = true>
    BIOIK_DEV double value(int k) const {
        return at<UNIFORM>(k, pb->ops[k].gene, pb->ops[k].span, pb->ops[k].clip_min, pb->ops[k].clip_max);
    }
    // the same with the op's numbers handed in: the chain walk asks for them in ONE burst of scalar loads with the joint's constants (fk_walk_n), and the
    // parents' entries are read from LDS before anything waits -- a lone wavefront (one pose per call, the stragglers of a batch) otherwise waits four times per joint
    static constexpr bool takes_op_numbers = true;
    template <bool UNIFORM = true>
    BIOIK_DEV double at(int k, int g, double span, double cmin, double cmax) const {
        BIOIK_FP_STRICT
        const double parent_gene = p0g[k];
        const double d0 = p0d[k], d1 = p1d[k];
        if (g < 0) return parent_gene;  // inactive op: the seed's value, carried by every elite
        const uint32_t word = rng_child_word(base, (uint32_t)(g + 1));
        double r = rng_gauss32(word);
        double f = mutation_rate * span;
        double gn = parent_gene;
        gn += r * f;
        double parent_gradient = d0 * (1.0 - fmix) + d1 * fmix;
        double g2 = parent_gradient * gradient_factor;
        gn += g2;
#if defined(BIOIK_CLAMP_LIBRARY)
        gn = fmin(fmax(gn, cmin), cmax);
#else
        gn = UNIFORM ? p_clamp_uniform(gn, cmin, cmax) : p_clamp(gn, cmin, cmax);
#endif
        return gn;
    }
    BIOIK_DEV double operator()(int k) const { return value<true>(k); }
};
template <class PB>
BIOIK_DEV ChildX<PB> make_child_x(PB pb, uint32_t key, uint32_t ctr1, uint32_t child_index, const double* p0g, const double* p0d, const double* p1d) {
    ChildX<PB> c;
    c.pb = pb, c.p0g = p0g, c.p0d = p0d, c.p1d = p1d;
    c.base = rng_child_base(rng_child_stream(key, ctr1), child_index);
    c.mutation_rate = rng_child_rate(c.base);
    c.fmix = (child_index % 2u == 0u) ? 0.2 : 0.0;
    c.gradient_factor = (double)(child_index % 3u);
    return c;
}
// The same child with the mix of the parents' momentum read from a table: `parent_gradient = d0 * (1 - fmix) + d1 * fmix` has TWO forms per op in a
// generation (child index even: fmix = 0.2, odd: 0), built once per species and generation by the very operations ChildX performs per gene and
// child (child_parent_gradient).  A lane that walks several children of a generation then reads one number instead of two and spends one FP64
// instruction (x gradient_factor) instead of four per gene and child; the child's accessor shrinks by five registers.  The table takes 2 M doubles
// of the species' OTHER elite buffer, which nobody reads or writes between the start of a generation and its winner copy.
BIOIK_DEV double child_parent_gradient(double d0, double d1, int parity) {  // parity: child_index % 2
    BIOIK_FP_STRICT
    const double fmix = parity == 0 ? 0.2 : 0.0;
    return d0 * (1.0 - fmix) + d1 * fmix;
}
template <class PB>
struct ChildT {
    PB pb;
    const double *p0g, *pgrow;  // LDS, op-indexed: genes of parent 0; the row of the table for this child's parity
    uint32_t base;
    double mutation_rate, gradient_factor;
    BIOIK_DEV double operator()(int k) const { return at(k, pb->ops[k].gene, pb->ops[k].span, pb->ops[k].clip_min, pb->ops[k].clip_max); }
    static constexpr bool takes_op_numbers = true;  // (ChildX::at)
    BIOIK_DEV double at(int k, int g, double span, double cmin, double cmax) const {
        BIOIK_FP_STRICT
        const double parent_gene = p0g[k];
        const double pg = pgrow[k];
        if (g < 0) return parent_gene;
        const uint32_t word = rng_child_word(base, (uint32_t)(g + 1));
        double r = rng_gauss32(word);
        double f = mutation_rate * span;
        double gn = parent_gene;
        gn += r * f;
        double g2 = pg * gradient_factor;
        gn += g2;
        gn = p_clamp_uniform(gn, cmin, cmax);
        return gn;
    }
};
// pgtable: LDS, [2][M] (row = child_index % 2)
template <class PB>
BIOIK_DEV ChildT<PB> make_child_t(PB pb, uint32_t key, uint32_t ctr1, uint32_t child_index, const double* p0g, const double* pgtable, int M) {
    ChildT<PB> c;
    c.pb = pb, c.p0g = p0g;
    c.pgrow = pgtable + (int)(child_index % 2u) * M;
    c.base = rng_child_base(rng_child_stream(key, ctr1), child_index);
    c.mutation_rate = rng_child_rate(c.base);
    c.gradient_factor = (double)(child_index % 3u);
    return c;
}
// the elite with ONE gene advanced by a step (the memetic phase's finite differences), read where it is needed
struct PerturbX {
    const double* el;
    int op;
    double step;
    const double* el2 = nullptr;  // (the line search's second support point, where the lanes share the terms of both: solve_body's secondary_shared)
    BIOIK_DEV double operator()(int k) const { return k == op ? el[k] + step : el[k]; }
};

template <class PB>
BIOIK_DEV void reproduce_child(PB pb, uint32_t key, uint32_t ctr1, uint32_t child_index, const double* p0g, const double* p0d, const double* p1d,
                               double* xo, int xs, double* go, int gs) {
    const uint32_t ci[1] = {child_index};
    double* const xc[1] = {xo};
    reproduce_children<1>(pb, key, ctr1, ci, p0g, p0d, p1d, xc, xs, go, gs);
}

// ---------------------------------------------------------------------------------------------------------
// success test of one tip / of the gene-only primary goals (problem.cpp:259-341)
// ---------------------------------------------------------------------------------------------------------
BIOIK_DEV void rot_from_quat(Q4 q, double* R) {  // KDL::Rotation::Quaternion
    double x = q.x, y = q.y, z = q.z, w = q.w;
    double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    R[0] = w2 + x2 - y2 - z2;
    R[1] = 2 * x * y - 2 * w * z;
    R[2] = 2 * x * z + 2 * w * y;
    R[3] = 2 * x * y + 2 * w * z;
    R[4] = w2 - x2 + y2 - z2;
    R[5] = 2 * y * z - 2 * w * x;
    R[6] = 2 * x * z - 2 * w * y;
    R[7] = 2 * y * z + 2 * w * x;
    R[8] = w2 - x2 - y2 + z2;
}
BIOIK_DEV V3 kdl_get_rot(const double* d) {  // KDL::Rotation::GetRot
    const double epsilon = 1e-6, epsilon2 = 1e-5;
    if ((fabs(d[1] - d[3]) < epsilon) && (fabs(d[2] - d[6]) < epsilon) && (fabs(d[5] - d[7]) < epsilon)) {
        if ((fabs(d[1] + d[3]) < epsilon2) && (fabs(d[2] + d[6]) < epsilon2) && (fabs(d[5] + d[7]) < epsilon2) && (fabs(d[0] + d[4] + d[8] - 3) < epsilon2))
            return v3(0.0, 0.0, 0.0);
        double angle = BIOIK_PI;
        double xx = (d[0] + 1) / 2, yy = (d[4] + 1) / 2, zz = (d[8] + 1) / 2;
        double xy = (d[1] + d[3]) / 4, xz = (d[2] + d[6]) / 4, yz = (d[5] + d[7]) / 4;
        double x, y, z;
        if ((xx > yy) && (xx > zz)) {
            x = sqrt(xx);
            y = xy / x;
            z = xz / x;
        } else if (yy > zz) {
            y = sqrt(yy);
            x = xy / y;
            z = yz / y;
        } else {
            z = sqrt(zz);
            x = xz / z;
            y = yz / z;
        }
        return v3(x * angle, y * angle, z * angle);
    }
    double f = (d[0] + d[4] + d[8] - 1) / 2;
    double x = (d[7] - d[5]), y = (d[2] - d[6]), z = (d[3] - d[1]);
    double n = sqrt(x * x + y * y + z * z);
    double angle = bioik_atan2(n / 2, f);
    return v3(x / n * angle, y / n * angle, z / n * angle);
}
// Twist(Ma^-1 * diff(pa,pb), Ma^-1 * diff(Ma,Mb)) of problem.cpp:281/300/321
BIOIK_DEV void pose_twist(const F7& fa, const F7& fb, double* t6) {
    double A[9], B[9], R[9];
    rot_from_quat(fa.q, A);
    rot_from_quat(fb.q, B);
    double dx = fb.p.x - fa.p.x, dy = fb.p.y - fa.p.y, dz = fb.p.z - fa.p.z;
    t6[0] = A[0] * dx + A[3] * dy + A[6] * dz;
    t6[1] = A[1] * dx + A[4] * dy + A[7] * dz;
    t6[2] = A[2] * dx + A[5] * dy + A[8] * dz;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = A[0 * 3 + i] * B[0 * 3 + j] + A[1 * 3 + i] * B[1 * 3 + j] + A[2 * 3 + i] * B[2 * 3 + j];
    V3 rv = kdl_get_rot(R);
    V3 w = v3(A[0] * rv.x + A[1] * rv.y + A[2] * rv.z, A[3] * rv.x + A[4] * rv.y + A[5] * rv.z, A[6] * rv.x + A[7] * rv.y + A[8] * rv.z);
    t6[3] = A[0] * w.x + A[3] * w.y + A[6] * w.z;
    t6[4] = A[1] * w.x + A[4] * w.y + A[7] * w.z;
    t6[5] = A[2] * w.x + A[5] * w.y + A[8] * w.z;
}
BIOIK_DEV double angle_shortest_path(Q4 a, Q4 b) {  // tf2::Quaternion::angleShortestPath
    double s = sqrt(qdot(a, a) * qdot(b, b));
    double d = qdot(a, b);
    if (d < 0) return clamped_acos(-d / s) * 2.0;
    return clamped_acos(d / s) * 2.0;
}

// position / orientation / pose goals (problem.cpp:270-323): one out-of-line copy of the KDL-equivalent twist arithmetic
// (sqrt, atan2, acos, divisions) for all the success-test sites; values in, value out
// (the goal's numbers travel as a pointer into the query's LDS parameters: 22 argument registers, no stack traffic)
BIOIK_CALL int check_frame_goal(int type, const lds_f64* P, F7 fb, double dpos, double drot, double dtwist) {
    bool ok = true;
    F7 fa = f7_identity();
    if (type == G_POSITION) fa.p = v3(P[0], P[1], P[2]);
    else if (type == G_ORIENTATION) fa.q = Q4{P[0], P[1], P[2], P[3]};
    else fa = F7{{P[0], P[1], P[2]}, {P[3], P[4], P[5], P[6]}};
    if (type == G_POSITION) {
        if (dpos != BIOIK_DBL_MAX) ok = ok && (sqrt(len2(fb.p - fa.p)) <= dpos);
        if (dtwist != BIOIK_DBL_MAX) {
            double tw[6];
            pose_twist(fa, fb, tw);
            for (int k = 0; k < 3; k++) ok = ok && (fabs(tw[k]) < dtwist);
        }
    } else if (type == G_ORIENTATION) {
        if (drot != BIOIK_DBL_MAX) ok = ok && (angle_shortest_path(fb.q, fa.q) * 180 / BIOIK_PI <= drot);
        if (dtwist != BIOIK_DBL_MAX) {
            double tw[6];
            pose_twist(fa, fb, tw);
            for (int k = 3; k < 6; k++) ok = ok && (fabs(tw[k]) < dtwist);
        }
    } else {
        if (dpos != BIOIK_DBL_MAX || drot != BIOIK_DBL_MAX) {
            ok = ok && (sqrt(len2(fb.p - fa.p)) <= dpos);
            ok = ok && (angle_shortest_path(fb.q, fa.q) * 180 / BIOIK_PI <= drot);
        }
#if !defined(BIOIK_NO_POSE_ONLY)
        // The linear part of the twist is the position error turned into the goal's frame: as long as the tip is further from the goal than
        // 2 dtwist, one of its components is at least 2 / sqrt(3) dtwist and the test below fails whatever the rotation -- which is the case in
        // every step but the last few of a solve; the twist (a frame change, two rotation matrices, an acos) is then not computed.
        // (a goal orientation that is not a unit quaternion scales the twist: no shortcut then)
        if (dtwist != BIOIK_DBL_MAX && dtwist < 1e100 && len2(fb.p - fa.p) > 4.0 * dtwist * dtwist && fabs(qdot(fa.q, fa.q) - 1.0) < 1e-6) return 0;
#endif
        if (dtwist != BIOIK_DBL_MAX) {
            double tw[6];
            pose_twist(fa, fb, tw);
            for (int k = 0; k < 6; k++) ok = ok && (fabs(tw[k]) < dtwist);
        }
    }
    return ok ? 1 : 0;
}

BIOIK_DEV bool check_goal(ProbPtr pb, int g, const F7& fb, const XV& x, const QueryCtx& qc, double dpos, double drot, double dtwist) {
    const int type = pb->primary[g].type;
    const double* P = qc.par + pb->primary[g].param_off;
    bool ok = true;
    if (type == G_POSITION || type == G_ORIENTATION || type == G_POSE) {
        ok = check_frame_goal(type, (const lds_f64*)P, fb, dpos, drot, dtwist) != 0;
    } else {
        double dmax = fmin(BIOIK_DBL_MAX, fmin(p_fresh(dpos), dtwist));  // (p_fresh: computed here, not in front of the step loop and carried through it)
        double d = goal_eval(pb, type, pb->primary[g].var_op, pb->primary[g].var_seed, P, fb, x, qc) * pb->primary[g].weight_sq;
        ok = ok && (d < dmax * dmax);
    }
    return ok;
}

// exact FK of one individual + primary fitness, optionally with Problem::checkSolutionActiveVariables
// (ik_base.h:203-207, ik_parallel.h:173-181).  It serves the once-per-step call sites (initial fitness, species
// ranking, wipe-out, success test).
// (kept inline: ROCm 7.2's gfx950 backend rejects the generic-pointer aperture test it emits for LDS pointers that
// cross a real call boundary — "V_CMP_NE_U32 0, $src_shared_base: Operand has incorrect register class")
#define BIOIK_NOINLINE BIOIK_DEV
struct FitCheck {
    double fitness;
    int ok;
};
template <int COOP = 0, class PB>
BIOIK_NOINLINE FitCheck exact_fitness_check(PB pb, XV x, QueryCtx qc, double* slots, double dpos, double drot, double dtwist, int do_check,
                                            const double* prefix = nullptr) {
    bool good = true;
    double sum = 0.0;
    V3 bal = v3(0.0, 0.0, 0.0);
    fk_walk<COOP>(pb, x, slots, nullptr, [&](int t, const F7& f) {
        sum = tip_goals(pb, t, f, x, qc, sum);
        balance_tip(pb, t, f, bal);
#if !defined(BIOIK_NO_POSE_ONLY)
        if (do_check && pb->tips[t].pose_off >= 0) {
            good = check_frame_goal(G_POSE, (const lds_f64*)(qc.par + pb->tips[t].pose_off), f, dpos, drot, dtwist) != 0 && good;
        } else
#endif
        if (do_check) {
            const int g0 = pb->tips[t].goal_first, g1 = g0 + pb->tips[t].goal_count;
            for (int g = g0; g < g1; g++) good = check_goal(pb, g, f, x, qc, dpos, drot, dtwist) && good;
        }
    }, prefix);
    sum = nonlink_primary(pb, x, qc, sum);
    sum += balance_cost(pb, bal, qc);
    if (do_check) {
        const F7 zero = F7{{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        for (int g = pb->n_link_primary; g < pb->n_primary; g++) good = check_goal(pb, g, zero, x, qc, dpos, drot, dtwist) && good;
        if constexpr (pb_flavour<PB>::general) {  // a goal type the success test does not know: weighted cost < min(dpos, dtwist)^2 (problem.cpp:327-334)
            const double dmax = fmin(BIOIK_DBL_MAX, fmin(dpos, dtwist));
            for (int g = 0; g < pb->n_balance; g++) good = (balance_goal_cost(pb, g, bal, qc) < dmax * dmax) && good;
        }
    }
    return FitCheck{sum, good ? 1 : 0};
}
