// stand-in for moveit::core::RobotModel & friends: a container filled from the flat model description of
// include/bioik_hip.h, exposing the accessors bio_ik calls with MoveIt's published meaning.
#pragma once
#include <Eigen/Dense>
#include <cmath>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
// stand-in for urdf_model: what BalanceGoal::describe reads (src/goal_types.cpp:236-247)
namespace urdf {
struct Vector3 {
    double x = 0, y = 0, z = 0;
};
struct Pose {
    Vector3 position;
};
struct Inertial {
    Pose origin;
    double mass = 0;
};
struct Link {
    std::shared_ptr<Inertial> inertial;
};
struct ModelInterface {
    std::map<std::string, std::shared_ptr<Link>> links_;
    std::shared_ptr<const Link> getLink(const std::string& name) const {
        auto it = links_.find(name);
        return it == links_.end() ? std::shared_ptr<const Link>() : std::shared_ptr<const Link>(it->second);
    }
};
typedef std::shared_ptr<ModelInterface> ModelInterfaceSharedPtr;
}  // namespace urdf
namespace moveit {
namespace core {
struct VariableBounds {
    double min_position_ = 0, max_position_ = 0;
    bool position_bounded_ = false;
    double min_velocity_ = 0, max_velocity_ = 0;
    bool velocity_bounded_ = false;
    double min_acceleration_ = 0, max_acceleration_ = 0;
    bool acceleration_bounded_ = false;
};
class LinkModel;
class JointModel {
public:
    enum JointType { UNKNOWN, REVOLUTE, PRISMATIC, PLANAR, FLOATING, FIXED };
    std::string name_;
    JointType type_ = FIXED;
    int joint_index_ = 0, first_variable_index_ = 0;
    std::vector<std::string> variable_names_;
    std::vector<VariableBounds> variable_bounds_;
    const LinkModel* parent_link_ = nullptr;
    const LinkModel* child_link_ = nullptr;
    const JointModel* mimic_ = nullptr;
    double mimic_factor_ = 1.0, mimic_offset_ = 0.0;
    virtual ~JointModel() {}
    const std::string& getName() const { return name_; }
    JointType getType() const { return type_; }
    size_t getJointIndex() const { return (size_t)joint_index_; }
    size_t getFirstVariableIndex() const { return (size_t)first_variable_index_; }
    size_t getVariableCount() const { return variable_names_.size(); }
    const std::vector<std::string>& getVariableNames() const { return variable_names_; }
    const std::vector<VariableBounds>& getVariableBounds() const { return variable_bounds_; }
    const LinkModel* getParentLinkModel() const { return parent_link_; }
    const LinkModel* getChildLinkModel() const { return child_link_; }
    const JointModel* getMimic() const { return mimic_; }
    double getMimicFactor() const { return mimic_factor_; }
    double getMimicOffset() const { return mimic_offset_; }
    virtual void computeTransform(const double* joint_values, Eigen::Isometry3d& transf) const {
        transf.t = Eigen::Vector3d(0, 0, 0);
        transf.r = Eigen::Quaterniond().toRotationMatrix();
        if (type_ == PLANAR) {  // Translation(x, y, 0) * AngleAxis(theta, Z)
            transf.t = Eigen::Vector3d(joint_values[0], joint_values[1], 0.0);
            transf.r = Eigen::Quaterniond(std::cos(joint_values[2] * 0.5), 0.0, 0.0, std::sin(joint_values[2] * 0.5)).toRotationMatrix();
        }
    }
};
class RevoluteJointModel : public JointModel {
public:
    Eigen::Vector3d axis_;
    const Eigen::Vector3d& getAxis() const { return axis_; }
};
class PrismaticJointModel : public JointModel {
public:
    Eigen::Vector3d axis_;
    const Eigen::Vector3d& getAxis() const { return axis_; }
};
class FloatingJointModel : public JointModel {};
class PlanarJointModel : public JointModel {};
class FixedJointModel : public JointModel {};
class LinkModel {
public:
    std::string name_;
    int link_index_ = 0;
    const JointModel* parent_joint_ = nullptr;
    const LinkModel* parent_link_ = nullptr;
    Eigen::Isometry3d joint_origin_transform_;
    const std::string& getName() const { return name_; }
    size_t getLinkIndex() const { return (size_t)link_index_; }
    const JointModel* getParentJointModel() const { return parent_joint_; }
    const LinkModel* getParentLinkModel() const { return parent_link_; }
    const Eigen::Isometry3d& getJointOriginTransform() const { return joint_origin_transform_; }
};
class RobotModel;
class JointModelGroup {
public:
    std::string name_;
    const RobotModel* parent_model_ = nullptr;
    const RobotModel& getParentModel() const { return *parent_model_; }
    std::vector<const JointModel*> active_joint_models_, joint_models_;
    std::vector<std::string> variable_names_, joint_model_names_;
    const std::string& getName() const { return name_; }
    const std::vector<const JointModel*>& getActiveJointModels() const { return active_joint_models_; }
    const std::vector<const JointModel*>& getJointModels() const { return joint_models_; }
    const std::vector<std::string>& getVariableNames() const { return variable_names_; }
    const std::vector<std::string>& getJointModelNames() const { return joint_model_names_; }
    const std::vector<std::string>& getActiveJointModelNames() const { return joint_model_names_; }
    size_t getVariableCount() const { return variable_names_.size(); }
};
class RobotModel {
public:
    std::vector<std::unique_ptr<JointModel>> joints_;
    std::vector<std::unique_ptr<LinkModel>> links_;
    std::vector<const JointModel*> joint_ptrs_, mimic_joints_, active_joints_, joint_of_variable_;
    std::vector<const LinkModel*> link_ptrs_;
    std::vector<std::string> link_names_, joint_names_, variable_names_;
    std::map<std::string, int> variable_index_;
    std::map<std::string, std::unique_ptr<JointModelGroup>> groups_;
    urdf::ModelInterfaceSharedPtr urdf_ = std::make_shared<urdf::ModelInterface>();
    const urdf::ModelInterfaceSharedPtr& getURDF() const { return urdf_; }
    const std::string& getName() const { static std::string n = "robot"; return n; }
    size_t getVariableCount() const { return variable_names_.size(); }
    size_t getJointModelCount() const { return joint_ptrs_.size(); }
    size_t getLinkModelCount() const { return link_ptrs_.size(); }
    const std::vector<const JointModel*>& getJointModels() const { return joint_ptrs_; }
    const std::vector<const JointModel*>& getActiveJointModels() const { return active_joints_; }
    const std::vector<const LinkModel*>& getLinkModels() const { return link_ptrs_; }
    const std::vector<const JointModel*>& getMimicJointModels() const { return mimic_joints_; }
    const std::vector<std::string>& getLinkModelNames() const { return link_names_; }
    const std::vector<std::string>& getJointModelNames() const { return joint_names_; }
    const std::vector<std::string>& getVariableNames() const { return variable_names_; }
    const JointModel* getJointModel(size_t i) const { return joint_ptrs_[i]; }
    const JointModel* getJointModel(int i) const { return joint_ptrs_[(size_t)i]; }
    const JointModel* getJointModel(const std::string& n) const {
        for (auto* j : joint_ptrs_)
            if (j->getName() == n) return j;
        return nullptr;
    }
    const LinkModel* getLinkModel(size_t i) const { return link_ptrs_[i]; }
    const LinkModel* getLinkModel(int i) const { return link_ptrs_[(size_t)i]; }
    const LinkModel* getLinkModel(const std::string& n) const {
        for (auto* l : link_ptrs_)
            if (l->getName() == n) return l;
        return nullptr;
    }
    bool hasLinkModel(const std::string& n) const { return getLinkModel(n) != nullptr; }
    const JointModel* getJointOfVariable(size_t i) const { return joint_of_variable_[i]; }
    const JointModel* getJointOfVariable(const std::string& n) const { return joint_of_variable_[(size_t)getVariableIndex(n)]; }
    int getVariableIndex(const std::string& n) const {
        auto it = variable_index_.find(n);
        if (it == variable_index_.end()) throw std::runtime_error("Variable '" + n + "' is not known to model");
        return it->second;
    }
    const VariableBounds& getVariableBounds(const std::string& n) const {
        int v = getVariableIndex(n);
        const JointModel* j = joint_of_variable_[(size_t)v];
        return j->variable_bounds_[(size_t)v - j->getFirstVariableIndex()];
    }
    void getVariableDefaultPositions(std::vector<double>& values) const {
        values.assign(getVariableCount(), 0.0);
        for (size_t v = 0; v < values.size(); v++) {
            const VariableBounds& b = getVariableBounds(variable_names_[v]);
            if (!(b.min_position_ <= 0.0 && 0.0 <= b.max_position_)) values[v] = 0.5 * (b.min_position_ + b.max_position_);
        }
    }
    const JointModelGroup* getJointModelGroup(const std::string& n) const {
        auto it = groups_.find(n);
        return it == groups_.end() ? nullptr : it->second.get();
    }
    void interpolate(const double* from, const double*, double, double* state) const {
        for (size_t i = 0; i < getVariableCount(); i++) state[i] = from[i];
    }
    void enforcePositionBounds(double*) const {}
};
typedef std::shared_ptr<RobotModel> RobotModelPtr;
typedef std::shared_ptr<const RobotModel> RobotModelConstPtr;
}  // namespace core
}  // namespace moveit
namespace robot_model = moveit::core;
