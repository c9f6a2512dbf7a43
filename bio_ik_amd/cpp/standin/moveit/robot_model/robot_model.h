// stand-in for moveit/robot_model/robot_model.h: the accessors of moveit::core::RobotModel / LinkModel / JointModel / JointModelGroup
// that the plugin boundary reads, with MoveIt's published meaning (see ../../README.md).  The add* builders are stand-in only: a real
// RobotModel is built from URDF + SRDF.
#pragma once
#include <Eigen/Geometry>
#include <cmath>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
// stand-in for urdf_model: the inertial of a link, which BalanceGoal reads through RobotModel::getURDF() (src/goal_types.cpp:236-247)
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
    std::string xml_;  // stand-in only: the robot description a RobotModel(urdf, srdf) is built from (moveit/rdf_loader/rdf_loader.h)
    std::map<std::string, std::shared_ptr<Link>> links_;
    std::shared_ptr<const Link> getLink(const std::string& name) const {
        auto it = links_.find(name);
        return it == links_.end() ? std::shared_ptr<const Link>() : std::shared_ptr<const Link>(it->second);
    }
};
typedef std::shared_ptr<ModelInterface> ModelInterfaceSharedPtr;
}  // namespace urdf
namespace srdf {
struct Model {
    std::string xml_;  // stand-in only: the semantic description (groups, end effectors)
};
typedef std::shared_ptr<Model> ModelSharedPtr;
}  // namespace srdf
namespace kinematics {
class KinematicsBase;
}
namespace moveit {
namespace core {
struct VariableBounds {
    double min_position_ = 0, max_position_ = 0;
    bool position_bounded_ = false;
    double min_velocity_ = 0, max_velocity_ = 0;
    bool velocity_bounded_ = false;
};
class LinkModel;
class JointModel {
public:
    enum JointType { UNKNOWN, REVOLUTE, PRISMATIC, PLANAR, FLOATING, FIXED };
    std::string name_;
    JointType type_ = FIXED;
    int first_variable_index_ = -1;
    bool continuous_ = false;
    Eigen::Vector3d axis_;
    std::vector<std::string> variable_names_;
    std::vector<VariableBounds> variable_bounds_;
    const LinkModel* child_link_ = nullptr;
    const JointModel* mimic_ = nullptr;
    double mimic_factor_ = 1.0, mimic_offset_ = 0.0;
    virtual ~JointModel() {}
    const std::string& getName() const { return name_; }
    JointType getType() const { return type_; }
    int getFirstVariableIndex() const { return first_variable_index_; }
    size_t getVariableCount() const { return variable_names_.size(); }
    const std::vector<std::string>& getVariableNames() const { return variable_names_; }
    const std::vector<VariableBounds>& getVariableBounds() const { return variable_bounds_; }
    const LinkModel* getChildLinkModel() const { return child_link_; }
    const JointModel* getMimic() const { return mimic_; }
    double getMimicFactor() const { return mimic_factor_; }
    double getMimicOffset() const { return mimic_offset_; }
};
class RevoluteJointModel : public JointModel {
public:
    const Eigen::Vector3d& getAxis() const { return axis_; }
    bool isContinuous() const { return continuous_; }
};
class PrismaticJointModel : public JointModel {
public:
    const Eigen::Vector3d& getAxis() const { return axis_; }
};
class FixedJointModel : public JointModel {};
class FloatingJointModel : public JointModel {};
class PlanarJointModel : public JointModel {};
class LinkModel {
public:
    std::string name_;
    int link_index_ = 0;
    const JointModel* parent_joint_ = nullptr;
    const LinkModel* parent_link_ = nullptr;
    Eigen::Isometry3d joint_origin_transform_;
    const std::string& getName() const { return name_; }
    int getLinkIndex() const { return link_index_; }
    const JointModel* getParentJointModel() const { return parent_joint_; }
    const LinkModel* getParentLinkModel() const { return parent_link_; }
    const Eigen::Isometry3d& getJointOriginTransform() const { return joint_origin_transform_; }
};
class RobotModel;
class JointModelGroup {
public:
    std::shared_ptr<kinematics::KinematicsBase> solver_instance_;  // what the kinematics plugin loader attaches to the group
    const std::shared_ptr<kinematics::KinematicsBase>& getSolverInstance() const { return solver_instance_; }
    void setSolverInstance(const std::shared_ptr<kinematics::KinematicsBase>& s) { solver_instance_ = s; }  // stand-in only
    const RobotModel* parent_model_ = nullptr;
    const RobotModel& getParentModel() const { return *parent_model_; }
    std::string name_;
    std::vector<const JointModel*> joint_models_, active_joint_models_;
    std::vector<std::string> end_effector_tips_;
    const std::string& getName() const { return name_; }
    const std::vector<const JointModel*>& getJointModels() const { return joint_models_; }
    const std::vector<const JointModel*>& getActiveJointModels() const { return active_joint_models_; }
    bool getEndEffectorTips(std::vector<std::string>& tips) const {  // SRDF end effectors attached to the group (none in the stand-in unless set)
        tips = end_effector_tips_;
        return true;
    }
};
class RobotModel {
    std::vector<std::unique_ptr<JointModel>> joints_;
    std::vector<std::unique_ptr<LinkModel>> links_;
    std::vector<const JointModel*> joint_ptrs_, mimic_joints_;
    std::vector<const LinkModel*> link_ptrs_;
    std::vector<std::string> variable_names_;
    std::vector<const JointModel*> joint_of_variable_;
    std::map<std::string, std::unique_ptr<JointModelGroup>> groups_;
    urdf::ModelInterfaceSharedPtr urdf_ = std::make_shared<urdf::ModelInterface>();

public:
    RobotModel() {}
    // from URDF + SRDF, as MoveIt builds it (the stand-in parses with bio_ik/urdf.h; defined in moveit/rdf_loader/rdf_loader.h)
    RobotModel(const urdf::ModelInterfaceSharedPtr& urdf_model, const srdf::ModelSharedPtr& srdf_model);
    const urdf::ModelInterfaceSharedPtr& getURDF() const { return urdf_; }
    void setInertial(const std::string& link, double mass, double cx, double cy, double cz) {  // stand-in only (a real model parses <inertial>)
        auto l = std::make_shared<urdf::Link>();
        l->inertial = std::make_shared<urdf::Inertial>();
        l->inertial->mass = mass, l->inertial->origin.position.x = cx, l->inertial->origin.position.y = cy, l->inertial->origin.position.z = cz;
        urdf_->links_[link] = l;
    }
    const std::string& getName() const {
        static const std::string n = "robot";
        return n;
    }
    size_t getVariableCount() const { return variable_names_.size(); }
    const std::vector<std::string>& getVariableNames() const { return variable_names_; }
    const std::vector<const LinkModel*>& getLinkModels() const { return link_ptrs_; }
    const std::vector<const JointModel*>& getJointModels() const { return joint_ptrs_; }
    const std::vector<const JointModel*>& getMimicJointModels() const { return mimic_joints_; }
    const JointModel* getJointModel(const std::string& n) const {
        for (auto* j : joint_ptrs_)
            if (j->getName() == n) return j;
        return nullptr;
    }
    const LinkModel* getLinkModel(const std::string& n) const {
        for (auto* l : link_ptrs_)
            if (l->getName() == n) return l;
        return nullptr;
    }
    bool hasLinkModel(const std::string& n) const { return getLinkModel(n) != nullptr; }
    int getVariableIndex(const std::string& n) const {
        for (size_t i = 0; i < variable_names_.size(); i++)
            if (variable_names_[i] == n) return (int)i;
        throw std::runtime_error("Variable '" + n + "' is not known to model '" + getName() + "'");
    }
    const JointModel* getJointOfVariable(int v) const { return joint_of_variable_[(size_t)v]; }
    const VariableBounds& getVariableBounds(const std::string& n) const {
        const int v = getVariableIndex(n);
        const JointModel* j = joint_of_variable_[(size_t)v];
        return j->variable_bounds_[(size_t)(v - j->getFirstVariableIndex())];
    }
    void getVariableDefaultPositions(std::vector<double>& values) const {  // 0 when inside the bounds, else their middle
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
    // clamp bounded variables, wrap continuous revolute joints into [-pi, pi] (RevoluteJointModel::enforcePositionBounds)
    void enforcePositionBounds(double* state) const {
        for (size_t v = 0; v < variable_names_.size(); v++) {
            const JointModel* j = joint_of_variable_[v];
            const VariableBounds& b = j->variable_bounds_[v - (size_t)j->getFirstVariableIndex()];
            if (j->getType() == JointModel::REVOLUTE && j->continuous_) {
                double& x = state[v];
                if (x <= -M_PI || x > M_PI) {
                    x = std::fmod(x, 2.0 * M_PI);
                    if (x <= -M_PI) x += 2.0 * M_PI;
                    else if (x > M_PI) x -= 2.0 * M_PI;
                }
            } else if (b.position_bounded_) {
                if (state[v] < b.min_position_) state[v] = b.min_position_;
                if (state[v] > b.max_position_) state[v] = b.max_position_;
            }
        }
    }

    // ---- stand-in only: builders (a real model comes from URDF + SRDF) ----
    // type: "fixed" | "revolute" | "continuous" | "prismatic" (URDF semantics); rpy as in URDF <origin>
    void addLink(const std::string& link, const std::string& parent, const std::string& joint, const std::string& type, double x, double y, double z,
                 double roll, double pitch, double yaw, double ax, double ay, double az, double lower = 0, double upper = 0, double velocity = 0) {
        const double cr = std::cos(roll / 2), sr = std::sin(roll / 2), cp = std::cos(pitch / 2), sp = std::sin(pitch / 2), cy = std::cos(yaw / 2), sy = std::sin(yaw / 2);
        Eigen::Isometry3d origin;
        origin.linear() = Eigen::Quaterniond(cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy).toRotationMatrix();
        origin.translation() = Eigen::Vector3d(x, y, z);
        addLink(link, parent, joint, type, origin, ax, ay, az, lower, upper, velocity);
    }
    void addLink(const std::string& link, const std::string& parent, const std::string& joint, const std::string& type, const Eigen::Isometry3d& origin,
                 double ax, double ay, double az, double lower = 0, double upper = 0, double velocity = 0) {
        JointModel* j;
        if (type == "fixed") j = new FixedJointModel(), j->type_ = JointModel::FIXED;
        else if (type == "prismatic") j = new PrismaticJointModel(), j->type_ = JointModel::PRISMATIC;
        else if (type == "revolute" || type == "continuous") j = new RevoluteJointModel(), j->type_ = JointModel::REVOLUTE, j->continuous_ = type == "continuous";
        else throw std::runtime_error("stand-in RobotModel: unsupported joint type " + type);
        j->name_ = joint;
        const double n = std::sqrt(ax * ax + ay * ay + az * az);
        j->axis_ = n > 0 ? Eigen::Vector3d(ax / n, ay / n, az / n) : Eigen::Vector3d(ax, ay, az);
        LinkModel* l = new LinkModel();
        l->name_ = link, l->link_index_ = (int)links_.size(), l->parent_joint_ = j;
        l->parent_link_ = parent.empty() ? nullptr : getLinkModel(parent);
        if (!parent.empty() && !l->parent_link_) throw std::runtime_error("stand-in RobotModel: unknown parent link " + parent);
        l->joint_origin_transform_ = origin;
        j->child_link_ = l;
        if (j->type_ != JointModel::FIXED) {
            j->first_variable_index_ = (int)variable_names_.size();
            j->variable_names_.push_back(joint);
            VariableBounds b;
            b.min_position_ = j->continuous_ ? -M_PI : lower, b.max_position_ = j->continuous_ ? M_PI : upper;
            b.position_bounded_ = !j->continuous_;
            b.max_velocity_ = velocity, b.min_velocity_ = -velocity, b.velocity_bounded_ = velocity > 0;
            j->variable_bounds_.push_back(b);
            variable_names_.push_back(joint);
            joint_of_variable_.push_back(j);
        }
        joints_.emplace_back(j), links_.emplace_back(l);
        joint_ptrs_.push_back(j), link_ptrs_.push_back(l);
    }
    void setMimic(const std::string& joint, const std::string& of, double factor, double offset) {
        for (auto& j : joints_)
            if (j->name_ == joint) {
                j->mimic_ = getJointModel(of), j->mimic_factor_ = factor, j->mimic_offset_ = offset;
                mimic_joints_.push_back(j.get());
            }
    }
    // a group from an explicit joint list (SRDF <group><joint .../>...) with its end-effector tips
    void addJointsGroup(const std::string& name, const std::vector<std::string>& joints, const std::vector<std::string>& tips) {
        std::unique_ptr<JointModelGroup> g(new JointModelGroup());
        g->name_ = name;
        for (auto& jn : joints) {
            const JointModel* j = getJointModel(jn);
            if (!j) throw std::runtime_error("stand-in RobotModel: unknown joint " + jn);
            g->joint_models_.push_back(j);
            if (j->getType() != JointModel::FIXED && !j->getMimic()) g->active_joint_models_.push_back(j);
        }
        g->end_effector_tips_ = tips;
        g->parent_model_ = this;
        groups_[name] = std::move(g);
    }
    JointModelGroup* getJointModelGroup(const std::string& name) {  // (non-const: the stand-in's loader attaches the solver instance)
        auto it = groups_.find(name);
        return it == groups_.end() ? nullptr : it->second.get();
    }
    void addChainGroup(const std::string& name, const std::string& base, const std::string& tip) {
        std::unique_ptr<JointModelGroup> g(new JointModelGroup());
        g->name_ = name;
        std::vector<const JointModel*> chain;
        for (const LinkModel* l = getLinkModel(tip); l && l->getName() != base; l = l->getParentLinkModel()) chain.insert(chain.begin(), l->getParentJointModel());
        for (auto* j : chain) {
            g->joint_models_.push_back(j);
            if (j->getType() != JointModel::FIXED && !j->getMimic()) g->active_joint_models_.push_back(j);
        }
        g->parent_model_ = this;
        groups_[name] = std::move(g);
    }
};
typedef std::shared_ptr<RobotModel> RobotModelPtr;
typedef std::shared_ptr<const RobotModel> RobotModelConstPtr;
}  // namespace core
}  // namespace moveit
namespace robot_model = moveit::core;
