// stand-in for moveit/kinematics_base/kinematics_base.h (MoveIt 1.x): the abstract interface pluginlib hands to MoveIt, with the
// virtuals and protected members the reference plugin overrides / uses (see ../../README.md)
#pragma once
#include <functional>
#include <string>
#include <vector>

#include <geometry_msgs/Pose.h>
#include <moveit/robot_model/robot_model.h>
#include <moveit/robot_state/robot_state.h>
#include <moveit_msgs/MoveItErrorCodes.h>
namespace kinematics {
struct KinematicsQueryOptions {  // (a plain struct in MoveIt: no virtual member, hence the reference's registry of BioIK option objects)
    bool lock_redundant_joints = false;
    bool return_approximate_solution = false;
};
class KinematicsBase {
public:
    typedef std::function<void(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_solution, moveit_msgs::MoveItErrorCodes& error_code)>
        IKCallbackFn;
    virtual ~KinematicsBase() {}
    virtual bool getPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, std::vector<double>& solution,
                               moveit_msgs::MoveItErrorCodes& error_code, const KinematicsQueryOptions& options = KinematicsQueryOptions()) const = 0;
    virtual bool searchPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, double timeout, std::vector<double>& solution,
                                  moveit_msgs::MoveItErrorCodes& error_code, const KinematicsQueryOptions& options = KinematicsQueryOptions()) const = 0;
    virtual bool searchPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, double timeout,
                                  const std::vector<double>& consistency_limits, std::vector<double>& solution, moveit_msgs::MoveItErrorCodes& error_code,
                                  const KinematicsQueryOptions& options = KinematicsQueryOptions()) const = 0;
    virtual bool searchPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, double timeout, std::vector<double>& solution,
                                  const IKCallbackFn& solution_callback, moveit_msgs::MoveItErrorCodes& error_code,
                                  const KinematicsQueryOptions& options = KinematicsQueryOptions()) const = 0;
    virtual bool searchPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, double timeout,
                                  const std::vector<double>& consistency_limits, std::vector<double>& solution, const IKCallbackFn& solution_callback,
                                  moveit_msgs::MoveItErrorCodes& error_code, const KinematicsQueryOptions& options = KinematicsQueryOptions()) const = 0;
    // multi-tip form (MoveIt's default forwards the first pose to the single-pose overload; bio_ik overrides it)
    virtual bool searchPositionIK(const std::vector<geometry_msgs::Pose>& ik_poses, const std::vector<double>& ik_seed_state, double timeout,
                                  const std::vector<double>& consistency_limits, std::vector<double>& solution, const IKCallbackFn& solution_callback,
                                  moveit_msgs::MoveItErrorCodes& error_code, const KinematicsQueryOptions& options = KinematicsQueryOptions(),
                                  const moveit::core::RobotState* /*context_state*/ = nullptr) const {
        if (ik_poses.size() != 1) return false;
        return searchPositionIK(ik_poses[0], ik_seed_state, timeout, consistency_limits, solution, solution_callback, error_code, options);
    }
    virtual bool getPositionFK(const std::vector<std::string>& link_names, const std::vector<double>& joint_angles, std::vector<geometry_msgs::Pose>& poses) const = 0;
    virtual bool initialize(const std::string& robot_description, const std::string& group_name, const std::string& base_frame, const std::string& tip_frame,
                            double search_discretization) = 0;
    virtual bool initialize(const std::string& robot_description, const std::string& group_name, const std::string& base_frame,
                            const std::vector<std::string>& tip_frames, double search_discretization) {
        return tip_frames.size() == 1 && initialize(robot_description, group_name, base_frame, tip_frames[0], search_discretization);
    }
    virtual bool initialize(const moveit::core::RobotModel& /*robot_model*/, const std::string& /*group_name*/, const std::string& /*base_frame*/,
                            const std::vector<std::string>& /*tip_frames*/, double /*search_discretization*/) {
        return false;
    }
    virtual const std::string& getGroupName() const { return group_name_; }
    virtual const std::string& getBaseFrame() const { return base_frame_; }
    virtual const std::vector<std::string>& getTipFrames() const { return tip_frames_; }
    virtual const std::vector<std::string>& getJointNames() const = 0;
    virtual const std::vector<std::string>& getLinkNames() const = 0;
    virtual bool supportsGroup(const moveit::core::JointModelGroup* jmg, std::string* error_text_out = nullptr) const = 0;
    void setValues(const std::string& robot_description, const std::string& group_name, const std::string& base_frame, const std::vector<std::string>& tip_frames,
                   double search_discretization) {
        robot_description_ = robot_description, group_name_ = group_name, base_frame_ = base_frame, tip_frames_ = tip_frames;
        search_discretization_ = search_discretization;
    }

protected:
    std::string robot_description_, group_name_, base_frame_;
    std::vector<std::string> tip_frames_;
    double search_discretization_ = 0.0;
};
}  // namespace kinematics

// RobotState::setFromIK (declared in the robot_state stand-in): MoveIt's flow reduced to what a kinematics plugin sees -- the seed is
// the state's own group variables in the solver's joint order, the poses go to the solver unchanged (callers of this stand-in give them
// in the solver's base frame), the state is the context state, and a solution is written back into the state.
inline bool moveit::core::RobotState::setFromIK(const JointModelGroup* group, const EigenSTL::vector_Affine3d& poses, const std::vector<std::string>& /*tips*/,
                                                unsigned int /*attempts*/, double timeout, const GroupStateValidityCallbackFn& /*constraint*/,
                                                const kinematics::KinematicsQueryOptions& options) {
    const std::shared_ptr<kinematics::KinematicsBase>& solver = group->getSolverInstance();
    if (!solver) return false;
    std::vector<double> seed, solution;
    for (auto& jn : solver->getJointNames()) seed.push_back(getVariablePosition(jn));
    std::vector<geometry_msgs::Pose> ik_poses;
    for (auto& T : poses) {
        const Eigen::Quaterniond q(T.rotation());
        geometry_msgs::Pose p;
        p.position.x = T.translation().x(), p.position.y = T.translation().y(), p.position.z = T.translation().z();
        p.orientation.x = q.x(), p.orientation.y = q.y(), p.orientation.z = q.z(), p.orientation.w = q.w();
        ik_poses.push_back(p);
    }
    moveit_msgs::MoveItErrorCodes error;
    if (!solver->searchPositionIK(ik_poses, seed, timeout, std::vector<double>(), solution, kinematics::KinematicsBase::IKCallbackFn(), error, options, this)) return false;
    for (size_t i = 0; i < solution.size(); i++) setVariablePosition(solver->getJointNames()[i], solution[i]);
    return true;
}
