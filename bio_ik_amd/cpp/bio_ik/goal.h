// bio_ik/goal.h — C++ host-side mirror of the reference's goal (cost) plugin interface
// (reference include/bio_ik/goal.h:49-129) for the MI355X build.
//
// Same class and member names as the reference.  Instead of a virtual `evaluate(const GoalContext&)` that the CPU
// solver calls per individual, a built-in goal serialises itself for the device: `gpuOpcode()` (BIOIK_GOAL_* of
// include/bioik_hip.h), `gpuLinkName()` / `gpuVariableName()` (what `describe()` puts into the GoalContext,
// goal.h:87-90) and `gpuParams()` (the per-query numbers).  Goals that need host callbacks return opcode -1 and make the
// plugin report BIOIK_ERR_UNSUPPORTED.
// tf2 / MoveIt are not required: positions and orientations are the small PODs below (x y z / x y z w).
#pragma once
#include <cmath>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../../include/bioik_hip.h"
#if defined(BIOIK_WITH_KINEMATICS_BASE)  // built inside the MoveIt plugin (src/kinematics_plugin_hip.cpp): MoveIt's own option struct
#include <moveit/kinematics_base/kinematics_base.h>
#endif

namespace bio_ik {

struct Vector3 {
    double x = 0, y = 0, z = 0;
    Vector3() {}
    Vector3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    Vector3 normalized() const {
        double l = std::sqrt(x * x + y * y + z * z);
        return Vector3(x / l, y / l, z / l);
    }
};
struct Quaternion {
    double x = 0, y = 0, z = 0, w = 1;
    Quaternion() {}
    Quaternion(double x_, double y_, double z_, double w_) : x(x_), y(y_), z(z_), w(w_) {}
    Quaternion normalized() const {
        double l = std::sqrt(x * x + y * y + z * z + w * w);
        return Quaternion(x / l, y / l, z / l, w / l);
    }
};

class Goal {  // reference goal.h:97-119
protected:
    bool secondary_ = false;
    double weight_ = 1.0;

public:
    virtual ~Goal() {}
    bool isSecondary() const { return secondary_; }
    double getWeight() const { return weight_; }
    void setWeight(double w) { weight_ = w; }
    // ---- device serialisation (replaces describe()/evaluate() on the GPU path) ----
    virtual int gpuOpcode() const { return -1; }
    virtual std::string gpuLinkName() const { return std::string(); }
    virtual std::string gpuVariableName() const { return std::string(); }
    virtual void gpuParams(std::vector<double>&) const {}
};

#if defined(BIOIK_WITH_KINEMATICS_BASE)
typedef kinematics::KinematicsQueryOptions KinematicsQueryOptions;
#else
// kinematics::KinematicsQueryOptions stand-in (moveit/kinematics_base/kinematics_base.h) for builds without MoveIt
struct KinematicsQueryOptions {
    bool lock_redundant_joints = false;
    bool return_approximate_solution = false;
    virtual ~KinematicsQueryOptions() {}
};
#endif

// MoveIt hands the options to the plugin as a `const kinematics::KinematicsQueryOptions&`, a struct without virtual members, so a
// BioIK options object is recognised by its address in a process-wide registry (reference src/kinematics_plugin.cpp:75-101)
inline std::mutex& bioIKKinematicsQueryOptionsMutex() {
    static std::mutex m;
    return m;
}
inline std::unordered_set<const void*>& bioIKKinematicsQueryOptionsList() {
    static std::unordered_set<const void*> l;
    return l;
}

// reference goal.h:121-129
struct BioIKKinematicsQueryOptions : KinematicsQueryOptions {
    std::vector<std::unique_ptr<Goal>> goals;
    std::vector<std::string> fixed_joints;
    bool replace = false;
    mutable double solution_fitness = 0;
    BioIKKinematicsQueryOptions() {
        std::lock_guard<std::mutex> lock(bioIKKinematicsQueryOptionsMutex());
        bioIKKinematicsQueryOptionsList().insert(this);
    }
    ~BioIKKinematicsQueryOptions() {
        std::lock_guard<std::mutex> lock(bioIKKinematicsQueryOptionsMutex());
        bioIKKinematicsQueryOptionsList().erase(this);
    }
};
inline const BioIKKinematicsQueryOptions* toBioIKKinematicsQueryOptions(const void* ptr) {  // kinematics_plugin.cpp:93-99
    std::lock_guard<std::mutex> lock(bioIKKinematicsQueryOptionsMutex());
    return bioIKKinematicsQueryOptionsList().count(ptr) ? (const BioIKKinematicsQueryOptions*)ptr : nullptr;
}

}  // namespace bio_ik
