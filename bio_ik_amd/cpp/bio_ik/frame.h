// bio_ik/frame.h — the vector / quaternion / frame types of the goal interface (reference include/bio_ik/frame.h:45-58:
// `typedef tf2::Quaternion Quaternion; typedef tf2::Vector3 Vector3; struct Frame { Vector3 pos; Quaternion rot; }`).
//
// In a ROS workspace the two typedefs ARE tf2's classes, exactly as in the reference, so user code that hands tf2 values to the goal
// setters compiles unchanged.  Where tf2 is not installed (this repository's build image) the two classes below provide the subset of
// tf2's interface that goals use — accessors x() y() z() w(), arithmetic, dot / cross / length / normalized, the axis-angle constructor —
// and convert implicitly from ANY type with those accessors (tf::Vector3, tf2::Vector3, Eigen::Vector3d, KDL ...), so the same user
// code compiles there too.  Host-side only: the solver's arithmetic lives in bio_ik_amd/csrc.
#pragma once
#include <cmath>

#if !defined(BIOIK_NO_TF2) && defined(__has_include)
#if __has_include(<tf2/LinearMath/Quaternion.h>) && __has_include(<tf2/LinearMath/Vector3.h>)
#define BIOIK_HAVE_TF2 1
#include <tf2/LinearMath/Quaternion.h>
#include <tf2/LinearMath/Vector3.h>
#endif
#endif

namespace bio_ik {

#if defined(BIOIK_HAVE_TF2)
typedef tf2::Vector3 Vector3;
typedef tf2::Quaternion Quaternion;
#else
class Vector3 {
    double v_[3];

public:
    Vector3() : v_{0, 0, 0} {}
    Vector3(double x, double y, double z) : v_{x, y, z} {}
    // anything with x() y() z(): tf::Vector3, tf2::Vector3, Eigen::Vector3d, ...
    template <class T, class = decltype(double(((const T*)nullptr)->x()) + double(((const T*)nullptr)->y()) + double(((const T*)nullptr)->z()))>
    Vector3(const T& o) : v_{double(o.x()), double(o.y()), double(o.z())} {}
    double x() const { return v_[0]; }
    double y() const { return v_[1]; }
    double z() const { return v_[2]; }
    void setX(double x) { v_[0] = x; }
    void setY(double y) { v_[1] = y; }
    void setZ(double z) { v_[2] = z; }
    void setValue(double x, double y, double z) { v_[0] = x, v_[1] = y, v_[2] = z; }
    double dot(const Vector3& o) const { return v_[0] * o.v_[0] + v_[1] * o.v_[1] + v_[2] * o.v_[2]; }
    Vector3 cross(const Vector3& o) const { return Vector3(v_[1] * o.v_[2] - v_[2] * o.v_[1], v_[2] * o.v_[0] - v_[0] * o.v_[2], v_[0] * o.v_[1] - v_[1] * o.v_[0]); }
    double length2() const { return dot(*this); }
    double length() const { return std::sqrt(length2()); }
    double distance2(const Vector3& o) const { return (o - *this).length2(); }
    double distance(const Vector3& o) const { return (o - *this).length(); }
    Vector3 normalized() const { return *this * (1.0 / length()); }  // tf2: v /= length() is v *= 1 / length()
    Vector3& normalize() { return *this = normalized(); }
    Vector3 operator+(const Vector3& o) const { return Vector3(v_[0] + o.v_[0], v_[1] + o.v_[1], v_[2] + o.v_[2]); }
    Vector3 operator-(const Vector3& o) const { return Vector3(v_[0] - o.v_[0], v_[1] - o.v_[1], v_[2] - o.v_[2]); }
    Vector3 operator-() const { return Vector3(-v_[0], -v_[1], -v_[2]); }
    Vector3 operator*(double s) const { return Vector3(v_[0] * s, v_[1] * s, v_[2] * s); }
    Vector3 operator/(double s) const { return *this * (1.0 / s); }
    Vector3& operator+=(const Vector3& o) { return *this = *this + o; }
    Vector3& operator-=(const Vector3& o) { return *this = *this - o; }
    Vector3& operator*=(double s) { return *this = *this * s; }
};
inline Vector3 operator*(double s, const Vector3& v) { return v * s; }

class Quaternion {
    double v_[4];

public:
    Quaternion() : v_{0, 0, 0, 1} {}
    Quaternion(double x, double y, double z, double w) : v_{x, y, z, w} {}
    Quaternion(const Vector3& axis, double angle) { setRotation(axis, angle); }  // tf2::Quaternion(axis, angle)
    template <class T, class = decltype(double(((const T*)nullptr)->x()) + double(((const T*)nullptr)->y()) + double(((const T*)nullptr)->z()) +
                                        double(((const T*)nullptr)->w()))>
    Quaternion(const T& o) : v_{double(o.x()), double(o.y()), double(o.z()), double(o.w())} {}
    double x() const { return v_[0]; }
    double y() const { return v_[1]; }
    double z() const { return v_[2]; }
    double w() const { return v_[3]; }
    void setValue(double x, double y, double z, double w) { v_[0] = x, v_[1] = y, v_[2] = z, v_[3] = w; }
    void setRotation(const Vector3& axis, double angle) {
        const double s = std::sin(angle * 0.5) / axis.length();
        setValue(axis.x() * s, axis.y() * s, axis.z() * s, std::cos(angle * 0.5));
    }
    double dot(const Quaternion& o) const { return v_[0] * o.v_[0] + v_[1] * o.v_[1] + v_[2] * o.v_[2] + v_[3] * o.v_[3]; }
    double length2() const { return dot(*this); }
    double length() const { return std::sqrt(length2()); }
    Quaternion normalized() const {
        const double s = 1.0 / length();  // tf2: q / length() is q * (1 / length())
        return Quaternion(v_[0] * s, v_[1] * s, v_[2] * s, v_[3] * s);
    }
    Quaternion& normalize() { return *this = normalized(); }
    Quaternion inverse() const { return Quaternion(-v_[0], -v_[1], -v_[2], v_[3]); }
    Quaternion operator*(const Quaternion& q) const {  // Hamilton product
        return Quaternion(v_[3] * q.v_[0] + v_[0] * q.v_[3] + v_[1] * q.v_[2] - v_[2] * q.v_[1], v_[3] * q.v_[1] + v_[1] * q.v_[3] + v_[2] * q.v_[0] - v_[0] * q.v_[2],
                          v_[3] * q.v_[2] + v_[2] * q.v_[3] + v_[0] * q.v_[1] - v_[1] * q.v_[0], v_[3] * q.v_[3] - v_[0] * q.v_[0] - v_[1] * q.v_[1] - v_[2] * q.v_[2]);
    }
    Quaternion operator+(const Quaternion& o) const { return Quaternion(v_[0] + o.v_[0], v_[1] + o.v_[1], v_[2] + o.v_[2], v_[3] + o.v_[3]); }
    Quaternion operator-(const Quaternion& o) const { return Quaternion(v_[0] - o.v_[0], v_[1] - o.v_[1], v_[2] - o.v_[2], v_[3] - o.v_[3]); }
    Quaternion operator-() const { return Quaternion(-v_[0], -v_[1], -v_[2], -v_[3]); }
    Quaternion operator*(double s) const { return Quaternion(v_[0] * s, v_[1] * s, v_[2] * s, v_[3] * s); }
};
#endif

// v rotated by the unit quaternion q (what the reference's quat_mul_vec computes, include/bio_ik/frame.h:108-149)
inline Vector3 quatRotate(const Quaternion& q, const Vector3& v) {
    const Vector3 u(q.x(), q.y(), q.z());
    const Vector3 t = u.cross(v) * 2.0;
    return v + t * q.w() + u.cross(t);
}

struct Frame {  // reference include/bio_ik/frame.h:51-91
    Vector3 pos;
    Quaternion rot;
    Frame() : pos(0, 0, 0), rot(0, 0, 0, 1) {}
    Frame(const Vector3& p, const Quaternion& r) : pos(p), rot(r) {}
    const Vector3& getPosition() const { return pos; }
    const Quaternion& getOrientation() const { return rot; }
    void setPosition(const Vector3& p) { pos = p; }
    void setOrientation(const Quaternion& q) { rot = q; }
    static const Frame& identity() {
        static const Frame f;
        return f;
    }
};

}  // namespace bio_ik
