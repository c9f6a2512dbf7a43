// stand-in for moveit/robot_state/robot_state.h: variable positions and global link transforms (see ../../README.md)
#pragma once
#include <functional>

#include <eigen_stl_containers/eigen_stl_containers.h>
#include <moveit/robot_model/robot_model.h>
namespace kinematics {
struct KinematicsQueryOptions;
}
namespace moveit {
namespace core {
class RobotState;
typedef std::function<bool(RobotState* robot_state, const JointModelGroup* joint_group, const double* joint_group_variable_values)> GroupStateValidityCallbackFn;
class RobotState {
    RobotModelConstPtr model_;
    std::vector<double> position_;

public:
    explicit RobotState(const RobotModelConstPtr& model) : model_(model), position_(model->getVariableCount(), 0.0) {}
    const RobotModelConstPtr& getRobotModel() const { return model_; }
    void setToDefaultValues() { model_->getVariableDefaultPositions(position_); }
    const double* getVariablePositions() const { return position_.data(); }
    double* getVariablePositions() { return position_.data(); }
    void setVariablePositions(const std::vector<double>& p) { position_ = p; }
    void update(bool /*force*/ = false) {}  // (MoveIt: refresh the cached link transforms; here they are computed where they are asked for)
    void setVariablePosition(const std::string& name, double v) { position_[(size_t)model_->getVariableIndex(name)] = v; }
    double getVariablePosition(const std::string& name) const { return position_[(size_t)model_->getVariableIndex(name)]; }
    // RobotState::setFromIK, the multi-tip overload with explicit options (defined with the KinematicsBase stand-in, kinematics_base.h):
    // the group's solver instance is asked for a solution seeded with this state, which is written back on success
    bool setFromIK(const JointModelGroup* group, const EigenSTL::vector_Affine3d& poses, const std::vector<std::string>& tips, unsigned int attempts,
                   double timeout, const GroupStateValidityCallbackFn& constraint, const kinematics::KinematicsQueryOptions& options);
    Eigen::Isometry3d getGlobalLinkTransform(const std::string& link_name) const {
        std::vector<const LinkModel*> chain;
        for (const LinkModel* l = model_->getLinkModel(link_name); l; l = l->getParentLinkModel()) chain.insert(chain.begin(), l);
        if (chain.empty()) throw std::runtime_error("RobotState: unknown link " + link_name);
        Eigen::Isometry3d T;
        for (const LinkModel* l : chain) {
            T = T * l->getJointOriginTransform();
            const JointModel* j = l->getParentJointModel();
            double v = 0.0;
            if (j->getVariableCount()) v = position_[(size_t)j->getFirstVariableIndex()];
            if (j->getMimic()) v = position_[(size_t)j->getMimic()->getFirstVariableIndex()] * j->getMimicFactor() + j->getMimicOffset();
            Eigen::Isometry3d J;
            if (j->getType() == JointModel::REVOLUTE) {
                const double s = std::sin(v / 2);
                J.linear() = Eigen::Quaterniond(std::cos(v / 2), j->axis_.x() * s, j->axis_.y() * s, j->axis_.z() * s).toRotationMatrix();
            } else if (j->getType() == JointModel::PRISMATIC) {
                J.translation() = Eigen::Vector3d(j->axis_.x() * v, j->axis_.y() * v, j->axis_.z() * v);
            }
            T = T * J;
        }
        return T;
    }
};
}  // namespace core
}  // namespace moveit
namespace robot_state = moveit::core;
