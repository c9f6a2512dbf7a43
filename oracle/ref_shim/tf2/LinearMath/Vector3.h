// stand-in for tf2/LinearMath/Vector3.h (ROS geometry2): the members bio_ik uses, published semantics.
#pragma once
#include <cmath>
namespace tf2 {
typedef double tf2Scalar;
inline double tf2Sqrt(double x) { return std::sqrt(x); }
inline double tf2Acos(double x) {
    if (x < -1.0) x = -1.0;
    if (x > 1.0) x = 1.0;
    return std::acos(x);
}
class Vector3 {
public:
    double m_floats[4];
    Vector3() {}
    Vector3(double x, double y, double z) { m_floats[0] = x, m_floats[1] = y, m_floats[2] = z, m_floats[3] = 0.0; }
    const double& x() const { return m_floats[0]; }
    const double& y() const { return m_floats[1]; }
    const double& z() const { return m_floats[2]; }
    const double& getX() const { return m_floats[0]; }
    const double& getY() const { return m_floats[1]; }
    const double& getZ() const { return m_floats[2]; }
    void setX(double v) { m_floats[0] = v; }
    void setY(double v) { m_floats[1] = v; }
    void setZ(double v) { m_floats[2] = v; }
    void setValue(double x, double y, double z) { m_floats[0] = x, m_floats[1] = y, m_floats[2] = z, m_floats[3] = 0.0; }
    double& operator[](int i) { return m_floats[i]; }
    const double& operator[](int i) const { return m_floats[i]; }
    Vector3& operator+=(const Vector3& v) { m_floats[0] += v.m_floats[0], m_floats[1] += v.m_floats[1], m_floats[2] += v.m_floats[2]; return *this; }
    Vector3& operator-=(const Vector3& v) { m_floats[0] -= v.m_floats[0], m_floats[1] -= v.m_floats[1], m_floats[2] -= v.m_floats[2]; return *this; }
    Vector3& operator*=(const double& s) { m_floats[0] *= s, m_floats[1] *= s, m_floats[2] *= s; return *this; }
    Vector3& operator/=(const double& s) { return *this *= 1.0 / s; }
    double dot(const Vector3& v) const { return m_floats[0] * v.m_floats[0] + m_floats[1] * v.m_floats[1] + m_floats[2] * v.m_floats[2]; }
    double length2() const { return dot(*this); }
    double length() const { return tf2Sqrt(length2()); }
    double distance2(const Vector3& v) const;
    double distance(const Vector3& v) const;
    Vector3& normalize() { return *this /= length(); }
    Vector3 normalized() const;
    double angle(const Vector3& v) const {
        double s = tf2Sqrt(length2() * v.length2());
        return tf2Acos(dot(v) / s);
    }
    Vector3 cross(const Vector3& v) const {
        return Vector3(m_floats[1] * v.m_floats[2] - m_floats[2] * v.m_floats[1], m_floats[2] * v.m_floats[0] - m_floats[0] * v.m_floats[2],
                       m_floats[0] * v.m_floats[1] - m_floats[1] * v.m_floats[0]);
    }
    bool operator==(const Vector3& o) const { return m_floats[0] == o.m_floats[0] && m_floats[1] == o.m_floats[1] && m_floats[2] == o.m_floats[2] && m_floats[3] == o.m_floats[3]; }
    bool operator!=(const Vector3& o) const { return !(*this == o); }
};
inline Vector3 operator+(const Vector3& a, const Vector3& b) { return Vector3(a.m_floats[0] + b.m_floats[0], a.m_floats[1] + b.m_floats[1], a.m_floats[2] + b.m_floats[2]); }
inline Vector3 operator-(const Vector3& a, const Vector3& b) { return Vector3(a.m_floats[0] - b.m_floats[0], a.m_floats[1] - b.m_floats[1], a.m_floats[2] - b.m_floats[2]); }
inline Vector3 operator-(const Vector3& a) { return Vector3(-a.m_floats[0], -a.m_floats[1], -a.m_floats[2]); }
inline Vector3 operator*(const Vector3& a, const double& s) { return Vector3(a.m_floats[0] * s, a.m_floats[1] * s, a.m_floats[2] * s); }
inline Vector3 operator*(const double& s, const Vector3& a) { return a * s; }
inline Vector3 operator/(const Vector3& a, const double& s) { return a * (1.0 / s); }
inline double Vector3::distance2(const Vector3& v) const { return (v - *this).length2(); }
inline double Vector3::distance(const Vector3& v) const { return (v - *this).length(); }
inline Vector3 Vector3::normalized() const { return *this / length(); }
}  // namespace tf2
