// stand-in for tf2/LinearMath/Quaternion.h (ROS geometry2): the members bio_ik uses, published semantics.
#pragma once
#include <cfloat>
#include "Vector3.h"
namespace tf2 {
class Quaternion {
public:
    double m_floats[4];
    Quaternion() {}
    Quaternion(double x, double y, double z, double w) { m_floats[0] = x, m_floats[1] = y, m_floats[2] = z, m_floats[3] = w; }
    Quaternion(const Vector3& axis, double angle) { setRotation(axis, angle); }
    const double& x() const { return m_floats[0]; }
    const double& y() const { return m_floats[1]; }
    const double& z() const { return m_floats[2]; }
    const double& w() const { return m_floats[3]; }
    const double& getX() const { return m_floats[0]; }
    const double& getY() const { return m_floats[1]; }
    const double& getZ() const { return m_floats[2]; }
    const double& getW() const { return m_floats[3]; }
    void setValue(double x, double y, double z, double w) { m_floats[0] = x, m_floats[1] = y, m_floats[2] = z, m_floats[3] = w; }
    void setX(double v) { m_floats[0] = v; }
    void setY(double v) { m_floats[1] = v; }
    void setZ(double v) { m_floats[2] = v; }
    void setW(double v) { m_floats[3] = v; }
    double& operator[](int i) { return m_floats[i]; }
    const double& operator[](int i) const { return m_floats[i]; }
    void setRotation(const Vector3& axis, double angle) {
        double d = axis.length();
        double s = std::sin(angle * 0.5) / d;
        setValue(axis.x() * s, axis.y() * s, axis.z() * s, std::cos(angle * 0.5));
    }
    Quaternion& operator+=(const Quaternion& q) { for (int i = 0; i < 4; i++) m_floats[i] += q.m_floats[i]; return *this; }
    Quaternion& operator-=(const Quaternion& q) { for (int i = 0; i < 4; i++) m_floats[i] -= q.m_floats[i]; return *this; }
    Quaternion& operator*=(const double& s) { for (int i = 0; i < 4; i++) m_floats[i] *= s; return *this; }
    Quaternion& operator/=(const double& s) { return *this *= 1.0 / s; }
    Quaternion& operator*=(const Quaternion& q) {
        setValue(m_floats[3] * q.x() + m_floats[0] * q.m_floats[3] + m_floats[1] * q.z() - m_floats[2] * q.y(),
                 m_floats[3] * q.y() + m_floats[1] * q.m_floats[3] + m_floats[2] * q.x() - m_floats[0] * q.z(),
                 m_floats[3] * q.z() + m_floats[2] * q.m_floats[3] + m_floats[0] * q.y() - m_floats[1] * q.x(),
                 m_floats[3] * q.m_floats[3] - m_floats[0] * q.x() - m_floats[1] * q.y() - m_floats[2] * q.z());
        return *this;
    }
    double dot(const Quaternion& q) const { return m_floats[0] * q.x() + m_floats[1] * q.y() + m_floats[2] * q.z() + m_floats[3] * q.m_floats[3]; }
    double length2() const { return dot(*this); }
    double length() const { return tf2Sqrt(length2()); }
    Quaternion& normalize() { return *this /= length(); }
    Quaternion operator*(const double& s) const { return Quaternion(x() * s, y() * s, z() * s, m_floats[3] * s); }
    Quaternion operator/(const double& s) const { return *this * (1.0 / s); }
    Quaternion normalized() const { return *this / length(); }
    double angleShortestPath(const Quaternion& q) const {
        double s = tf2Sqrt(length2() * q.length2());
        if (dot(q) < 0) return tf2Acos(dot(-q) / s) * 2.0;
        return tf2Acos(dot(q) / s) * 2.0;
    }
    double getAngle() const { return 2.0 * tf2Acos(m_floats[3]); }
    Vector3 getAxis() const {
        double s_squared = 1.0 - m_floats[3] * m_floats[3];
        if (s_squared < 10.0 * DBL_EPSILON) return Vector3(1.0, 0.0, 0.0);
        double s = 1.0 / tf2Sqrt(s_squared);
        return Vector3(m_floats[0] * s, m_floats[1] * s, m_floats[2] * s);
    }
    Quaternion inverse() const { return Quaternion(-m_floats[0], -m_floats[1], -m_floats[2], m_floats[3]); }
    Quaternion operator+(const Quaternion& q2) const { return Quaternion(x() + q2.x(), y() + q2.y(), z() + q2.z(), m_floats[3] + q2.m_floats[3]); }
    Quaternion operator-(const Quaternion& q2) const { return Quaternion(x() - q2.x(), y() - q2.y(), z() - q2.z(), m_floats[3] - q2.m_floats[3]); }
    Quaternion operator-() const { return Quaternion(-x(), -y(), -z(), -m_floats[3]); }
    static const Quaternion& getIdentity() {
        static const Quaternion identityQuat(0.0, 0.0, 0.0, 1.0);
        return identityQuat;
    }
};
inline Quaternion operator*(const Quaternion& q1, const Quaternion& q2) {
    return Quaternion(q1.w() * q2.x() + q1.x() * q2.w() + q1.y() * q2.z() - q1.z() * q2.y(), q1.w() * q2.y() + q1.y() * q2.w() + q1.z() * q2.x() - q1.x() * q2.z(),
                      q1.w() * q2.z() + q1.z() * q2.w() + q1.x() * q2.y() - q1.y() * q2.x(), q1.w() * q2.w() - q1.x() * q2.x() - q1.y() * q2.y() - q1.z() * q2.z());
}
inline Quaternion inverse(const Quaternion& q) { return q.inverse(); }
}  // namespace tf2
