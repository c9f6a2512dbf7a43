#pragma once
// MoveIt built against FCL >= 0.6: bio_ik compiles its TouchGoal out (goal_types.h:330, problem.h:65)
#define FCL_VERSION_CHECK(major, minor, patch) ((major) * 10000 + (minor) * 100 + (patch))
#define MOVEIT_FCL_VERSION FCL_VERSION_CHECK(0, 6, 0)
#include <memory>
namespace fcl {
struct Vec3f { double v[3]; };
}
namespace collision_detection {
struct FCLGeometry {};
typedef std::shared_ptr<const FCLGeometry> FCLGeometryConstPtr;
}  // namespace collision_detection
