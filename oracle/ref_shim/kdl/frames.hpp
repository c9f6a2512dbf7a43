// stand-in for orocos-kdl frames.hpp: Vector, Rotation, Frame, Twist, diff, Equal — published semantics (frames.cpp/.inl)
#pragma once
#include <cmath>
namespace KDL {
class Vector {
public:
    double data[3];
    Vector() { data[0] = data[1] = data[2] = 0.0; }
    Vector(double x, double y, double z) { data[0] = x, data[1] = y, data[2] = z; }
    double x() const { return data[0]; }
    double y() const { return data[1]; }
    double z() const { return data[2]; }
    void x(double v) { data[0] = v; }
    void y(double v) { data[1] = v; }
    void z(double v) { data[2] = v; }
    double operator()(int i) const { return data[i]; }
    double& operator()(int i) { return data[i]; }
    double operator[](int i) const { return data[i]; }
    double& operator[](int i) { return data[i]; }
    double Norm() const { return std::sqrt(data[0] * data[0] + data[1] * data[1] + data[2] * data[2]); }
    static Vector Zero() { return Vector(0, 0, 0); }
};
inline Vector operator+(const Vector& a, const Vector& b) { return Vector(a.data[0] + b.data[0], a.data[1] + b.data[1], a.data[2] + b.data[2]); }
inline Vector operator-(const Vector& a, const Vector& b) { return Vector(a.data[0] - b.data[0], a.data[1] - b.data[1], a.data[2] - b.data[2]); }
inline Vector operator*(const Vector& a, double s) { return Vector(a.data[0] * s, a.data[1] * s, a.data[2] * s); }
inline Vector operator*(double s, const Vector& a) { return a * s; }
inline Vector operator/(const Vector& a, double s) { return Vector(a.data[0] / s, a.data[1] / s, a.data[2] / s); }
class Rotation {
public:
    double data[9];
    Rotation() { for (int i = 0; i < 9; i++) data[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    Rotation(double Xx, double Yx, double Zx, double Xy, double Yy, double Zy, double Xz, double Yz, double Zz) {
        data[0] = Xx, data[1] = Yx, data[2] = Zx, data[3] = Xy, data[4] = Yy, data[5] = Zy, data[6] = Xz, data[7] = Yz, data[8] = Zz;
    }
    static Rotation Identity() { return Rotation(); }
    static Rotation Quaternion(double x, double y, double z, double w) {
        double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
        return Rotation(w2 + x2 - y2 - z2, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y, 2 * x * y + 2 * w * z, w2 - x2 + y2 - z2, 2 * y * z - 2 * w * x,
                        2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, w2 - x2 - y2 + z2);
    }
    void GetQuaternion(double& x, double& y, double& z, double& w) const {
        double trace = data[0] + data[4] + data[8];
        double epsilon = 1E-12;
        if (trace > epsilon) {
            double s = 0.5 / std::sqrt(trace + 1.0);
            w = 0.25 / s, x = (data[7] - data[5]) * s, y = (data[2] - data[6]) * s, z = (data[3] - data[1]) * s;
        } else if (data[0] > data[4] && data[0] > data[8]) {
            double s = 2.0 * std::sqrt(1.0 + data[0] - data[4] - data[8]);
            w = (data[7] - data[5]) / s, x = 0.25 * s, y = (data[1] + data[3]) / s, z = (data[2] + data[6]) / s;
        } else if (data[4] > data[8]) {
            double s = 2.0 * std::sqrt(1.0 + data[4] - data[0] - data[8]);
            w = (data[2] - data[6]) / s, x = (data[1] + data[3]) / s, y = 0.25 * s, z = (data[5] + data[7]) / s;
        } else {
            double s = 2.0 * std::sqrt(1.0 + data[8] - data[0] - data[4]);
            w = (data[3] - data[1]) / s, x = (data[2] + data[6]) / s, y = (data[5] + data[7]) / s, z = 0.25 * s;
        }
    }
    Rotation Inverse() const { return Rotation(data[0], data[3], data[6], data[1], data[4], data[7], data[2], data[5], data[8]); }
    Vector operator*(const Vector& v) const {
        return Vector(data[0] * v.data[0] + data[1] * v.data[1] + data[2] * v.data[2], data[3] * v.data[0] + data[4] * v.data[1] + data[5] * v.data[2],
                      data[6] * v.data[0] + data[7] * v.data[1] + data[8] * v.data[2]);
    }
    double GetRotAngle(Vector& axis, double eps = 1e-6) const {
        double angle, x, y, z;
        double epsilon = eps, epsilon2 = eps * 10;
        if ((std::fabs(data[1] - data[3]) < epsilon) && (std::fabs(data[2] - data[6]) < epsilon) && (std::fabs(data[5] - data[7]) < epsilon)) {
            if ((std::fabs(data[1] + data[3]) < epsilon2) && (std::fabs(data[2] + data[6]) < epsilon2) && (std::fabs(data[5] + data[7]) < epsilon2) &&
                (std::fabs(data[0] + data[4] + data[8] - 3) < epsilon2)) {
                axis = Vector(0, 0, 1);
                return 0.0;
            }
            angle = M_PI;
            double xx = (data[0] + 1) / 2, yy = (data[4] + 1) / 2, zz = (data[8] + 1) / 2;
            double xy = (data[1] + data[3]) / 4, xz = (data[2] + data[6]) / 4, yz = (data[5] + data[7]) / 4;
            if ((xx > yy) && (xx > zz)) {
                x = std::sqrt(xx), y = xy / x, z = xz / x;
            } else if (yy > zz) {
                y = std::sqrt(yy), x = xy / y, z = yz / y;
            } else {
                z = std::sqrt(zz), x = xz / z, y = yz / z;
            }
            axis = Vector(x, y, z);
            return angle;
        }
        double f = (data[0] + data[4] + data[8] - 1) / 2;
        x = (data[7] - data[5]), y = (data[2] - data[6]), z = (data[3] - data[1]);
        axis = Vector(x, y, z);
        angle = std::atan2(axis.Norm() / 2, f);
        axis = axis / axis.Norm();
        return angle;
    }
    Vector GetRot() const {
        Vector axis;
        double angle = GetRotAngle(axis);
        return axis * angle;
    }
};
inline Rotation operator*(const Rotation& a, const Rotation& b) {
    Rotation r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.data[i * 3 + j] = a.data[i * 3 + 0] * b.data[0 * 3 + j] + a.data[i * 3 + 1] * b.data[1 * 3 + j] + a.data[i * 3 + 2] * b.data[2 * 3 + j];
    return r;
}
class Frame {
public:
    Vector p;
    Rotation M;
    Frame() {}
    Frame(const Rotation& R, const Vector& V) : p(V), M(R) {}
};
class Twist {
public:
    Vector vel, rot;
    Twist() {}
    Twist(const Vector& v, const Vector& r) : vel(v), rot(r) {}
    static Twist Zero() { return Twist(); }
    double operator()(int i) const { return i < 3 ? vel(i) : rot(i - 3); }
    double& operator()(int i) { return i < 3 ? vel(i) : rot(i - 3); }
};
inline Vector diff(const Vector& a, const Vector& b, double dt = 1) { return (b - a) / dt; }
inline Vector diff(const Rotation& R_a_b1, const Rotation& R_a_b2, double dt = 1) {
    Rotation R_b1_b2(R_a_b1.Inverse() * R_a_b2);
    return R_a_b1 * R_b1_b2.GetRot() / dt;
}
inline bool Equal(double a, double b, double eps = 1e-6) { return std::fabs(a - b) < eps; }  // utility.h: (tmp < eps) && (tmp > -eps)
inline bool Equal(const Vector& a, const Vector& b, double eps = 1e-6) { return Equal(a.data[0], b.data[0], eps) && Equal(a.data[1], b.data[1], eps) && Equal(a.data[2], b.data[2], eps); }
inline bool Equal(const Twist& a, const Twist& b, double eps = 1e-6) { return Equal(a.rot, b.rot, eps) && Equal(a.vel, b.vel, eps); }
}  // namespace KDL
