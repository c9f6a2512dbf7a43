// bio_ik/goal.h — the goal (cost) plugin interface of the MI355X build: source compatible with the reference's
// include/bio_ik/goal.h:46-129.
//
// `GoalContext`, `Goal::describe(GoalContext&)` and `Goal::evaluate(const GoalContext&)` have the reference's names, signatures and
// meaning, so goal classes written against the reference — its built-in ones and a user's own subclasses — compile unchanged.
// What is new is the second face of a goal: a built-in goal also SERIALISES itself for the device — `gpuOpcode()` (BIOIK_GOAL_* of
// include/bioik_hip.h), `gpuLinkName()` / `gpuVariableName()` (what `describe()` adds to the context, goal.h:87-90) and `gpuParams()`
// (its per-query numbers) — because on the MI355X the costs of a whole population are evaluated inside the solver kernels, not by a
// virtual call per individual.  `evaluate()` remains the definition of the cost (the parity tests hold the kernels against it) and is
// what runs for goals on the host (bio_ik/goal_eval.h).  A goal without a device opcode (a user subclass, JointFunctionGoal,
// LinkFunctionGoal) cannot steer the device search: it takes part through the hybrid path of plugin_core.h -- the device searches over the goals it can
// evaluate and returns several candidates per query, the host scores them with ALL goals and picks.
#pragma once
#include <sys/types.h>

#include <cmath>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

#include <bioik_hip.h>
#if defined(BIOIK_WITH_KINEMATICS_BASE)  // built inside the MoveIt plugin (src/kinematics_plugin_hip.cpp): MoveIt's own option struct
#include <moveit/kinematics_base/kinematics_base.h>
#include <moveit/robot_model/joint_model_group.h>
#include <moveit/robot_model/robot_model.h>
#endif

#include "frame.h"
#include "robot_info.h"

namespace bio_ik {

class HostGoalProblem;

class GoalContext {  // reference goal.h:49-95
protected:
    const double* active_variable_positions_ = nullptr;
    const Frame* tip_link_frames_ = nullptr;
    std::vector<ssize_t> goal_variable_indices_;
    std::vector<size_t> goal_link_indices_;
    bool goal_secondary_ = false;
    std::vector<std::string> goal_link_names_, goal_variable_names_;
    double goal_weight_ = 1;
#if defined(BIOIK_WITH_KINEMATICS_BASE)
    const moveit::core::JointModelGroup* joint_model_group_ = nullptr;
#endif
    std::vector<size_t> problem_active_variables_;
    std::vector<size_t> problem_tip_link_indices_;
    std::vector<double> initial_guess_;
    std::vector<double> velocity_weights_;
    const RobotInfo* robot_info_ = nullptr;
    mutable std::vector<double> temp_vector_;

public:
    GoalContext() {}
    const Frame& getLinkFrame(size_t i = 0) const { return tip_link_frames_[goal_link_indices_[i]]; }
    double getVariablePosition(size_t i = 0) const {
        const ssize_t j = goal_variable_indices_[i];
        return j >= 0 ? active_variable_positions_[j] : initial_guess_[-1 - j];  // a fixed variable keeps the seed's value (goal.h:70-77)
    }
    const Frame& getProblemLinkFrame(size_t i) const { return tip_link_frames_[i]; }
    size_t getProblemLinkCount() const { return problem_tip_link_indices_.size(); }
    size_t getProblemLinkIndex(size_t i) const { return problem_tip_link_indices_[i]; }
    double getProblemVariablePosition(size_t i) const { return active_variable_positions_[i]; }
    size_t getProblemVariableCount() const { return problem_active_variables_.size(); }
    size_t getProblemVariableIndex(size_t i) const { return problem_active_variables_[i]; }
    double getProblemVariableInitialGuess(size_t i) const { return initial_guess_[problem_active_variables_[i]]; }
    double getProblemVariableWeight(size_t i) const { return velocity_weights_[i]; }
    const RobotInfo& getRobotInfo() const { return *robot_info_; }
    void addLink(const std::string& name) { goal_link_names_.push_back(name); }
    void addVariable(const std::string& name) { goal_variable_names_.push_back(name); }
    void setSecondary(bool secondary) { goal_secondary_ = secondary; }
    void setWeight(double weight) { goal_weight_ = weight; }
#if defined(BIOIK_WITH_KINEMATICS_BASE)
    const moveit::core::JointModelGroup& getJointModelGroup() const { return *joint_model_group_; }
    const moveit::core::RobotModel& getRobotModel() const { return joint_model_group_->getParentModel(); }
#endif
    std::vector<double>& getTempVector() const { return temp_vector_; }
    friend class HostGoalProblem;  // bio_ik/goal_eval.h: the host-side counterpart of the reference's `Problem`
};

class Goal {  // reference goal.h:97-119
protected:
    bool secondary_;
    double weight_;

public:
    Goal() : secondary_(false), weight_(1) {}
    virtual ~Goal() {}
    bool isSecondary() const { return secondary_; }
    double getWeight() const { return weight_; }
    void setWeight(double w) { weight_ = w; }
    virtual void describe(GoalContext& context) const {
        context.setSecondary(secondary_);
        context.setWeight(weight_);
    }
    virtual double evaluate(const GoalContext&) const { return 0; }
    // ---- device serialisation (what the solver kernels evaluate; -1: the goal exists on the host only) ----
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
