// stand-in: geometry_msgs PODs + tf2::fromMsg for quaternions
#pragma once
#include <tf2/LinearMath/Quaternion.h>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
}  // namespace geometry_msgs
namespace tf2 {
inline void fromMsg(const geometry_msgs::Quaternion& in, Quaternion& out) { out = Quaternion(in.x, in.y, in.z, in.w); }
}
