// Known answers for the stand-in third-party headers of oracle/ref_shim (tf2 LinearMath, orocos-KDL frames): the reference's own sources
// are compiled against these to pin the CPU restatement, so "restatement == reference" must not be satisfiable by a bug the two share in
// a helper.  Every expected value below is derived by hand from the libraries' PUBLISHED definitions (tf2/LinearMath/Quaternion.h,
// Vector3.h; orocos_kdl frames.cpp / frames.inl / utility.h) or from the mathematics of rotations -- never from the stand-ins themselves.
#include <cfloat>
#include <cmath>
#include <cstdio>

#include <kdl/frames.hpp>
#include <tf2/LinearMath/Quaternion.h>
#include <tf2/LinearMath/Vector3.h>

static int failures = 0;
#define EXPECT(c)                                                        \
    do {                                                                 \
        if (!(c)) {                                                      \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);  \
            failures++;                                                  \
        }                                                                \
    } while (0)
static bool near(double a, double b, double tol = 1e-15) { return std::fabs(a - b) <= tol; }

int main() {
    const double pi = 3.14159265358979323846, r2 = std::sqrt(0.5);
    // ---- tf2::Vector3 / Quaternion ------------------------------------------------------------------------------------------------
    {
        // normalisation multiplies by the reciprocal of the length (operator/ is `*this * (1 / s)`): 3 * (1/5) is 0.6000000000000001, 3/5 is 0.6
        const tf2::Vector3 v = tf2::Vector3(0, 3, 4).normalized();
        EXPECT(v.y() == 3.0 * (1.0 / 5.0) && v.y() != 3.0 / 5.0 && v.z() == 4.0 * (1.0 / 5.0));
        const tf2::Quaternion q = tf2::Quaternion(0, 0, 3, 4).normalized();
        EXPECT(q.z() == 3.0 * (1.0 / 5.0) && q.w() == 4.0 * (1.0 / 5.0));
        EXPECT(tf2::Vector3(1, 2, 3).cross(tf2::Vector3(4, 5, 6)) == tf2::Vector3(-3, 6, -3));
        EXPECT(tf2::Vector3(1, 2, 3).dot(tf2::Vector3(4, 5, 6)) == 32 && tf2::Vector3(1, 2, 2).length() == 3 && tf2::Vector3(1, 2, 3).distance2(tf2::Vector3(2, 4, 6)) == 14);
        EXPECT(near(tf2::Vector3(1, 0, 0).angle(tf2::Vector3(0, 2, 0)), pi / 2));
    }
    {
        // axis-angle constructor: (axis / |axis|) sin(angle / 2), cos(angle / 2); the axis need not be a unit vector
        const tf2::Quaternion q(tf2::Vector3(0, 0, 2), pi / 2);
        EXPECT(q.x() == 0 && q.y() == 0 && near(q.z(), r2) && near(q.w(), r2));
        // Hamilton product: a quarter turn about z twice is a half turn about z; i * j = k
        const tf2::Quaternion h = q * q;
        EXPECT(near(h.z(), 1.0) && near(h.w(), 0.0) && near(h.x(), 0) && near(h.y(), 0));
        const tf2::Quaternion k = tf2::Quaternion(1, 0, 0, 0) * tf2::Quaternion(0, 1, 0, 0);
        EXPECT(k.x() == 0 && k.y() == 0 && k.z() == 1 && k.w() == 0);
        EXPECT(q.inverse().z() == -q.z() && q.inverse().w() == q.w());
        // angleShortestPath: the rotation angle between two orientations, q and -q being the same orientation
        const tf2::Quaternion id(0, 0, 0, 1), a3(tf2::Vector3(1, 2, 2), 3.0), a4(tf2::Vector3(1, 2, 2), 4.0);
        EXPECT(near(id.angleShortestPath(a3), 3.0, 1e-14) && near(a3.angleShortestPath(id), 3.0, 1e-14));
        EXPECT(near(id.angleShortestPath(-a3), 3.0, 1e-14));
        EXPECT(near(id.angleShortestPath(a4), 2 * pi - 4.0, 1e-14));  // beyond pi the other way round is shorter
        EXPECT(near((a3 * 2.0).angleShortestPath(id * 3.0), 3.0, 1e-14));  // normalises by the two lengths
        EXPECT(id.angleShortestPath(id) == 0.0);
        // getAngle = 2 acos(w), clamped; getAxis falls back to (1, 0, 0) when 1 - w^2 < 10 epsilon
        EXPECT(near(a3.getAngle(), 3.0, 1e-14) && near(a3.getAxis().x(), 1.0 / 3, 1e-15) && near(a3.getAxis().z(), 2.0 / 3, 1e-15));
        EXPECT(tf2::Quaternion(0, 1e-9, 0, 1.0).getAxis() == tf2::Vector3(1, 0, 0));                 // 1 - w^2 = 0 < 10 eps
        EXPECT(tf2::Quaternion(0, 3e-8, 0, std::sqrt(1 - 9e-16)).getAxis() == tf2::Vector3(1, 0, 0));  // 9e-16 < 2.2e-15
        const tf2::Vector3 ax = tf2::Quaternion(tf2::Vector3(0, 1, 0), 1e-6).getAxis();                 // 1 - w^2 = 2.5e-13: a real axis
        EXPECT(near(ax.y(), 1.0, 1e-4) && ax.x() == 0 && ax.z() == 0);
        EXPECT(tf2::Quaternion(0, 0, 0, 1.5).getAngle() == 0.0);  // tf2Acos clamps its argument
    }
    // ---- KDL::Rotation ----------------------------------------------------------------------------------------------------------------
    {
        // Quaternion(x, y, z, w) -> matrix, row-major data[]: a quarter turn about z maps x to y
        const KDL::Rotation Rz = KDL::Rotation::Quaternion(0, 0, r2, r2);
        const KDL::Vector y = Rz * KDL::Vector(1, 0, 0);
        EXPECT(near(y.x(), 0) && near(y.y(), 1) && near(y.z(), 0));
        EXPECT(near(Rz.data[1], -1) && near(Rz.data[3], 1) && near(Rz.data[8], 1));
        const KDL::Vector back = Rz.Inverse() * y;
        EXPECT(near(back.x(), 1) && near(back.y(), 0));
        // GetQuaternion, its four branches: trace > 0; then the largest diagonal element decides (half turns about x, y, z have trace -1)
        double x, yy, z, w;
        KDL::Rotation::Identity().GetQuaternion(x, yy, z, w);
        EXPECT(x == 0 && yy == 0 && z == 0 && w == 1);
        Rz.GetQuaternion(x, yy, z, w);
        EXPECT(near(x, 0) && near(yy, 0) && near(z, r2) && near(w, r2));
        KDL::Rotation(1, 0, 0, 0, -1, 0, 0, 0, -1).GetQuaternion(x, yy, z, w);
        EXPECT(x == 1 && yy == 0 && z == 0 && w == 0);
        KDL::Rotation(-1, 0, 0, 0, 1, 0, 0, 0, -1).GetQuaternion(x, yy, z, w);
        EXPECT(x == 0 && yy == 1 && z == 0 && w == 0);
        KDL::Rotation(-1, 0, 0, 0, -1, 0, 0, 0, 1).GetQuaternion(x, yy, z, w);
        EXPECT(x == 0 && yy == 0 && z == 1 && w == 0);
        // 170 degrees about (0.6, 0.8, 0): trace = 1 + 2 cos < 0, branch "Yy largest"; the quaternion comes back up to its sign
        const double h = 170.0 * pi / 180 / 2;
        KDL::Rotation::Quaternion(0.6 * std::sin(h), 0.8 * std::sin(h), 0, std::cos(h)).GetQuaternion(x, yy, z, w);
        EXPECT(near(x, 0.6 * std::sin(h), 1e-15) && near(yy, 0.8 * std::sin(h), 1e-15) && near(z, 0, 1e-15) && near(w, std::cos(h), 1e-15));
    }
    {
        // GetRot / diff: the rotation vector.  Ordinary angles: axis * angle.
        const auto Rzq = [](double t) { return KDL::Rotation::Quaternion(0, 0, std::sin(t / 2), std::cos(t / 2)); };
        KDL::Vector d = KDL::diff(KDL::Rotation::Identity(), Rzq(0.3));
        EXPECT(near(d.x(), 0) && near(d.y(), 0) && near(d.z(), 0.3, 1e-15));
        d = KDL::diff(Rzq(0.2), Rzq(0.5));  // expressed in the frame both rotations are given in
        EXPECT(near(d.z(), 0.3, 1e-15) && near(d.x(), 0) && near(d.y(), 0));
        d = KDL::diff(KDL::Rotation::Identity(), Rzq(0.3), 0.1);
        EXPECT(near(d.z(), 3.0, 1e-14));
        // a dead zone around the identity: when the antisymmetric part is below 1e-6 and the symmetric part that of the identity, GetRot is zero
        d = KDL::diff(KDL::Rotation::Identity(), Rzq(1e-9));
        EXPECT(d.x() == 0 && d.y() == 0 && d.z() == 0);
        // half turns: the antisymmetric part vanishes, the axis comes from the diagonal, its largest component taken positive, angle = pi exactly
        const KDL::Rotation half(2 * 0.36 - 1, 2 * 0.48, 0, 2 * 0.48, 2 * 0.64 - 1, 0, 0, 0, -1);  // 2 a a^T - I, a = (0.6, 0.8, 0)
        d = half.GetRot();
        EXPECT(near(d.x(), 0.6 * pi, 1e-15) && near(d.y(), 0.8 * pi, 1e-15) && d.z() == 0);
        // ... and so do rotations within 1e-6 of a half turn (the snap of frames.cpp: angle = pi, not pi - 1e-9)
        d = KDL::diff(KDL::Rotation::Identity(), Rzq(pi - 1e-9));
        EXPECT(d.z() == pi && near(d.x(), 0, 1e-9) && near(d.y(), 0, 1e-9));
        d = KDL::diff(KDL::Rotation::Identity(), Rzq(pi - 1e-3));  // outside the snap: atan2 form
        EXPECT(near(d.z(), pi - 1e-3, 1e-12));
        const KDL::Vector v = KDL::diff(KDL::Vector(1, 2, 3), KDL::Vector(2, 4, 6), 0.5);
        EXPECT(v.x() == 2 && v.y() == 4 && v.z() == 6);
    }
    {
        // Equal: |a - b| < eps, strictly (utility.h), component-wise for vectors and twists; Twist(i): 0..2 velocity, 3..5 rotation
        EXPECT(KDL::Equal(1.0, 1.0 + 9e-7) && !KDL::Equal(1.0, 1.0 + 1e-6 + 1e-12) && !KDL::Equal(0.0, 1e-6, 1e-6) && KDL::Equal(0.0, 0.5, 1.0));
        KDL::Twist t(KDL::Vector(1, 2, 3), KDL::Vector(4, 5, 6));
        EXPECT(t(0) == 1 && t(2) == 3 && t(3) == 4 && t(5) == 6);
        EXPECT(KDL::Equal(t, KDL::Twist(KDL::Vector(1, 2, 3 + 5e-7), KDL::Vector(4, 5, 6))) && !KDL::Equal(t, KDL::Twist(KDL::Vector(1, 2, 3), KDL::Vector(4, 5 + 2e-6, 6))));
    }
    if (failures) return 1;
    std::printf("ok\n");
    return 0;
}
