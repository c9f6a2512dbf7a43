// stand-in for the slice of tf (tf/LinearMath) that user code of the bio_ik examples touches: tf::Vector3 and tf::Quaternion with
// accessors, arithmetic and the axis-angle constructor (see ../README.md).  Deliberately NOT the types of bio_ik/frame.h: what the
// goal setters accept from foreign vector types is part of what the tests check.
#pragma once
#include <cmath>
namespace tf {
class Vector3 {
    double m[3];

public:
    Vector3() : m{0, 0, 0} {}
    Vector3(double x, double y, double z) : m{x, y, z} {}
    double x() const { return m[0]; }
    double y() const { return m[1]; }
    double z() const { return m[2]; }
    Vector3 operator+(const Vector3& o) const { return Vector3(m[0] + o.m[0], m[1] + o.m[1], m[2] + o.m[2]); }
    Vector3 operator-(const Vector3& o) const { return Vector3(m[0] - o.m[0], m[1] - o.m[1], m[2] - o.m[2]); }
    Vector3 operator*(double s) const { return Vector3(m[0] * s, m[1] * s, m[2] * s); }
    double length() const { return std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]); }
};
class Quaternion {
    double m[4];

public:
    Quaternion() : m{0, 0, 0, 1} {}
    Quaternion(double x, double y, double z, double w) : m{x, y, z, w} {}
    Quaternion(const Vector3& axis, double angle) {
        const double s = std::sin(angle * 0.5) / axis.length();
        m[0] = axis.x() * s, m[1] = axis.y() * s, m[2] = axis.z() * s, m[3] = std::cos(angle * 0.5);
    }
    double x() const { return m[0]; }
    double y() const { return m[1]; }
    double z() const { return m[2]; }
    double w() const { return m[3]; }
};
}  // namespace tf
