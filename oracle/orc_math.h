// orc_math.h — ORACLE (test infrastructure): L1 math restated from the reference.
//
// Follows reference include/bio_ik/frame.h:51-259 expression by expression (same association order,
// so that a strict-IEEE build reproduces the reference's roundings) plus the handful of tf2 / KDL helper
// semantics the hot path touches.  tf2 (ROS geometry2, unpinned in reference package.xml:32-33) and
// orocos_kdl (unpinned, via tf2_kdl) are NOT on disk; their published algorithms are restated here and
// named at each function.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "../bio_ik_amd/csrc/bioik_fused.h"
#include "../bio_ik_amd/csrc/bioik_acos.h"

namespace orc {

// Arithmetic of the oracle.  0 (default): the reference's own expressions — libm sin/cos (forward_kinematics.h:96-97) and
// separate multiplies and adds (frame.h); in this mode the restatement is pinned bit-for-bit against the reference's code
// (tests/test_oracle_vs_reference.py).  1: "device arithmetic" — the bit-reproducible bioik_sincos and the fused forms of
// bioik_fused.h, both shared with the gfx950 kernels.  libm vs the GPU math library and fused vs unfused products differ in
// the last ulp, which bio2_memetic's line search amplifies into different trajectories, so the bit-for-bit parity tests of
// the kernels run the oracle in mode 1.  The two modes agree to a few ulp per operation.
inline int& trig_mode() {
    static int mode = 0;
    return mode;
}
#if defined(ORC_FIXED_ARITH_MODE)
// timing build (liboracle_ref.so): the mode is a compile-time constant, so the reference-arithmetic code carries no run-time
// test per multiply-add and vectorises as the reference's own code does
inline constexpr bool fused() { return ORC_FIXED_ARITH_MODE == 1; }
#else
inline bool fused() { return trig_mode() == 1; }
#endif
inline double madd(double a, double b, double c) { return fused() ? BK_FMA(a, b, c) : a * b + c; }  // a*b + c

struct Vec3 {
    double x, y, z;
};
struct Quat {
    double x, y, z, w;
};
// reference frame.h:51-56 (Vector3 is 4 doubles in tf2, the 4th is padding; not modelled)
struct Frame {
    Vec3 pos;
    Quat rot;
};

inline Frame identity_frame() { return Frame{{0, 0, 0}, {0, 0, 0, 1}}; }

// ---- tf2::Vector3 semantics (tf2/LinearMath/Vector3.h) ----
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(const Vec3& a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(const Vec3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(const Vec3& a, const Vec3& b) {
    if (fused()) return bk_dot3(a.x, a.y, a.z, b.x, b.y, b.z);
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
inline double length2(const Vec3& a) { return dot(a, a); }
inline double length(const Vec3& a) { return std::sqrt(length2(a)); }
inline double distance2(const Vec3& a, const Vec3& b) { return length2(b - a); }  // tf2: (v - *this).length2()
inline double distance(const Vec3& a, const Vec3& b) { return length(b - a); }
inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline Vec3 normalized(const Vec3& a) {  // tf2: *this / length(), and tf2's operator/ multiplies by the reciprocal
    double inv = 1.0 / length(a);
    return {a.x * inv, a.y * inv, a.z * inv};
}
inline double tf2_acos(double x) {  // tf2Acos clamps its argument (tf2/LinearMath/Scalar.h)
    if (x < -1.0) x = -1.0;
    if (x > 1.0) x = 1.0;
    return fused() ? bioik_acos(x) : std::acos(x);  // (device-arithmetic mode: the implementation the kernels share, bioik_acos.h)
}
inline double angle(const Vec3& a, const Vec3& b) {  // tf2::Vector3::angle
    double s = std::sqrt(length2(a) * length2(b));
    return tf2_acos(dot(a, b) / s);
}

// ---- tf2::Quaternion semantics (tf2/LinearMath/Quaternion.h) ----
inline double dot(const Quat& a, const Quat& b) {
    if (fused()) return bk_dot4(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
inline double length2(const Quat& a) { return dot(a, a); }
inline Quat operator+(const Quat& a, const Quat& b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline Quat operator-(const Quat& a, const Quat& b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline Quat operator-(const Quat& a) { return {-a.x, -a.y, -a.z, -a.w}; }
inline Quat normalized(const Quat& a) {  // tf2: *this / length() = *this * (1 / length())
    double inv = 1.0 / std::sqrt(length2(a));
    return {a.x * inv, a.y * inv, a.z * inv, a.w * inv};
}
inline Quat inverse(const Quat& q) { return {-q.x, -q.y, -q.z, q.w}; }  // tf2::Quaternion::inverse
// tf2 operator*(Quaternion, Quaternion): Hamilton product
inline Quat tf2_mul(const Quat& q1, const Quat& q2) {
    return {q1.w * q2.x + q1.x * q2.w + q1.y * q2.z - q1.z * q2.y,
            q1.w * q2.y + q1.y * q2.w + q1.z * q2.x - q1.x * q2.z,
            q1.w * q2.z + q1.z * q2.w + q1.x * q2.y - q1.y * q2.x,
            q1.w * q2.w - q1.x * q2.x - q1.y * q2.y - q1.z * q2.z};
}
inline double angle_shortest_path(const Quat& a, const Quat& b) {  // tf2::Quaternion::angleShortestPath
    double s = std::sqrt(length2(a) * length2(b));
    if (dot(a, b) < 0) return tf2_acos(dot(a, -b) / s) * 2.0;
    return tf2_acos(dot(a, b) / s) * 2.0;
}
inline double get_angle(const Quat& q) { return 2.0 * tf2_acos(q.w); }  // tf2::Quaternion::getAngle
inline Vec3 get_axis(const Quat& q) {                                   // tf2::Quaternion::getAxis
    double s_squared = 1.0 - q.w * q.w;
    if (s_squared < 10.0 * DBL_EPSILON) return {1.0, 0.0, 0.0};
    double s = 1.0 / std::sqrt(s_squared);
    return {q.x * s, q.y * s, q.z * s};
}

// ---- reference frame.h:108-149 ----
inline void quat_mul_vec(const Quat& q, const Vec3& v, Vec3& r) {
    if (fused()) {
        Vec3 o;
        bk_qrot(q.x, q.y, q.z, q.w, v.x, v.y, v.z, o.x, o.y, o.z);
        r = o;
        return;
    }
    double v_x = v.x, v_y = v.y, v_z = v.z;
    double q_x = q.x, q_y = q.y, q_z = q.z, q_w = q.w;
    if ((v_x == 0 && v_y == 0 && v_z == 0) || (q_x == 0 && q_y == 0 && q_z == 0 && q_w == 1)) {
        r = v;  // frame.h:122-126 short-circuit
        return;
    }
    double t_x = q_y * v_z - q_z * v_y;
    double t_y = q_z * v_x - q_x * v_z;
    double t_z = q_x * v_y - q_y * v_x;
    double r_x = q_w * t_x + q_y * t_z - q_z * t_y;
    double r_y = q_w * t_y + q_z * t_x - q_x * t_z;
    double r_z = q_w * t_z + q_x * t_y - q_y * t_x;
    r_x += r_x;
    r_y += r_y;
    r_z += r_z;
    r_x += v_x;
    r_y += v_y;
    r_z += v_z;
    r = {r_x, r_y, r_z};
}

// ---- reference frame.h:151-172 ----
inline void quat_mul_quat(const Quat& p, const Quat& q, Quat& r) {
    if (fused()) {
        Quat o;
        bk_qmul(p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w, o.x, o.y, o.z, o.w);
        r = o;
        return;
    }
    double p_x = p.x, p_y = p.y, p_z = p.z, p_w = p.w;
    double q_x = q.x, q_y = q.y, q_z = q.z, q_w = q.w;
    double r_x = (p_w * q_x + p_x * q_w) + (p_y * q_z - p_z * q_y);
    double r_y = (p_w * q_y - p_x * q_z) + (p_y * q_w + p_z * q_x);
    double r_z = (p_w * q_z + p_x * q_y) - (p_y * q_x - p_z * q_w);
    double r_w = (p_w * q_w - p_x * q_x) - (p_y * q_y + p_z * q_z);
    r = {r_x, r_y, r_z, r_w};
}

// ---- reference frame.h:174-187 ----
inline void concat(const Frame& a, const Frame& b, Frame& r) {
    Vec3 d;
    quat_mul_vec(a.rot, b.pos, d);
    Vec3 p = a.pos + d;
    Quat q;
    quat_mul_quat(a.rot, b.rot, q);
    r.pos = p;
    r.rot = q;
}
inline void concat(const Frame& a, const Frame& b, const Frame& c, Frame& r) {
    Frame tmp;
    concat(a, b, tmp);
    concat(tmp, c, r);
}
// ---- reference frame.h:189-209 ----
inline void invert(const Frame& a, Frame& r) {
    Quat qi = inverse(a.rot);
    Vec3 p;
    quat_mul_vec(qi, -a.pos, p);
    r.rot = qi;
    r.pos = p;
}
inline void change(const Frame& a, const Frame& b, const Frame& c, Frame& r) {
    Frame tmp;
    invert(b, tmp);
    concat(a, tmp, c, r);
}
// ---- reference frame.h:231-238 ----
inline void normalize_fast(Quat& q) {
    double f = (3.0 - length2(q)) * 0.5;
    q.x *= f;
    q.y *= f;
    q.z *= f;
    q.w *= f;
}
// ---- reference frame.h:240-259 (KDL::Twist as 6 doubles vel,rot) ----
inline void frame_twist(const Frame& a, const Frame& b, double* t6) {
    Frame ai, f;
    invert(a, ai);
    concat(ai, b, f);
    t6[0] = f.pos.x;
    t6[1] = f.pos.y;
    t6[2] = f.pos.z;
    double ra = get_angle(f.rot);
    if (ra > +M_PI) ra -= 2 * M_PI;
    Vec3 r = get_axis(f.rot) * ra;
    t6[3] = r.x;
    t6[4] = r.y;
    t6[5] = r.z;
}

// ---- orocos_kdl semantics used by problem.cpp:278-282, 297-301, 318-322 ----
// KDL::Rotation::Quaternion(x,y,z,w) (frames.cpp) -> row-major 3x3
inline void kdl_rotation_from_quat(const Quat& q, double* R) {
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
// KDL::Rotation::GetRot() = axis*angle via GetRotAngle(axis, epsilon=1e-6) (orocos_kdl >= 1.4 frames.cpp)
inline Vec3 kdl_get_rot(const double* d) {
    const double epsilon = 1e-6, epsilon2 = 1e-5;
    if ((std::fabs(d[1] - d[3]) < epsilon) && (std::fabs(d[2] - d[6]) < epsilon) && (std::fabs(d[5] - d[7]) < epsilon)) {
        if ((std::fabs(d[1] + d[3]) < epsilon2) && (std::fabs(d[2] + d[6]) < epsilon2) &&
            (std::fabs(d[5] + d[7]) < epsilon2) && (std::fabs(d[0] + d[4] + d[8] - 3) < epsilon2)) {
            return {0, 0, 0};  // identity: angle 0
        }
        // angle = PI
        double angle = M_PI;
        double xx = (d[0] + 1) / 2, yy = (d[4] + 1) / 2, zz = (d[8] + 1) / 2;
        double xy = (d[1] + d[3]) / 4, xz = (d[2] + d[6]) / 4, yz = (d[5] + d[7]) / 4;
        double x, y, z;
        if ((xx > yy) && (xx > zz)) {
            x = std::sqrt(xx);
            y = xy / x;
            z = xz / x;
        } else if (yy > zz) {
            y = std::sqrt(yy);
            x = xy / y;
            z = yz / y;
        } else {
            z = std::sqrt(zz);
            x = xz / z;
            y = yz / z;
        }
        return {x * angle, y * angle, z * angle};
    }
    double f = (d[0] + d[4] + d[8] - 1) / 2;
    double x = (d[7] - d[5]), y = (d[2] - d[6]), z = (d[3] - d[1]);
    double n = std::sqrt(x * x + y * y + z * z);
    double angle = fused() ? bioik_atan2(n / 2, f) : std::atan2(n / 2, f);
    return {x / n * angle, y / n * angle, z / n * angle};
}
// Twist( Ma^-1 * diff(pa,pb), Ma^-1 * diff(Ma,Mb) ), problem.cpp:281/300/321.
// KDL::diff(R_a, R_b) = R_a * (R_a^T R_b).GetRot(); so Ma^-1*diff = (Ma^T Mb).GetRot().
inline void kdl_pose_twist(const Frame& fa, const Frame& fb, double* t6) {
    double A[9], B[9];
    kdl_rotation_from_quat(fa.rot, A);
    kdl_rotation_from_quat(fb.rot, B);
    double dx = fb.pos.x - fa.pos.x, dy = fb.pos.y - fa.pos.y, dz = fb.pos.z - fa.pos.z;
    // A^T * d
    t6[0] = A[0] * dx + A[3] * dy + A[6] * dz;
    t6[1] = A[1] * dx + A[4] * dy + A[7] * dz;
    t6[2] = A[2] * dx + A[5] * dy + A[8] * dz;
    double R[9];  // A^T * B
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = A[0 * 3 + i] * B[0 * 3 + j] + A[1 * 3 + i] * B[1 * 3 + j] + A[2 * 3 + i] * B[2 * 3 + j];
    Vec3 rv = kdl_get_rot(R);
    // diff(Ma,Mb) = Ma * rv ; then Ma^-1 * that = rv (up to rounding; evaluated explicitly like KDL does)
    Vec3 w = {A[0] * rv.x + A[1] * rv.y + A[2] * rv.z, A[3] * rv.x + A[4] * rv.y + A[5] * rv.z,
              A[6] * rv.x + A[7] * rv.y + A[8] * rv.z};
    t6[3] = A[0] * w.x + A[3] * w.y + A[6] * w.z;
    t6[4] = A[1] * w.x + A[4] * w.y + A[7] * w.z;
    t6[5] = A[2] * w.x + A[5] * w.y + A[8] * w.z;
}

// ---- reference src/utils.h:319-333 ----
inline double mix(double a, double b, double f) { return a * (1.0 - f) + b * f; }
inline double clamp(double v, double lo, double hi) {
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

inline Frame frame_from7(const double* p) { return Frame{{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}}; }
inline void frame_to7(const Frame& f, double* p) {
    p[0] = f.pos.x;
    p[1] = f.pos.y;
    p[2] = f.pos.z;
    p[3] = f.rot.x;
    p[4] = f.rot.y;
    p[5] = f.rot.z;
    p[6] = f.rot.w;
}

}  // namespace orc
