// bio_ik/goal_types.h — the reference's built-in goal classes (include/bio_ik/goal_types.h:56-712) with the same
// names, constructors, setters and normalisation behaviour, serialising to the device opcodes of include/bioik_hip.h.
#pragma once
#include "goal.h"

namespace bio_ik {

class LinkGoalBase : public Goal {  // goal_types.h:56-78
    std::string link_name_;

public:
    LinkGoalBase() { weight_ = 1; }
    LinkGoalBase(const std::string& link_name, double weight) : link_name_(link_name) { weight_ = weight; }
    void setLinkName(const std::string& n) { link_name_ = n; }
    const std::string& getLinkName() const { return link_name_; }
    std::string gpuLinkName() const override { return link_name_; }
};

class PositionGoal : public LinkGoalBase {  // :80-97
    Vector3 position_;

public:
    PositionGoal() {}
    PositionGoal(const std::string& link_name, const Vector3& position, double weight = 1.0) : LinkGoalBase(link_name, weight), position_(position) {}
    const Vector3& getPosition() const { return position_; }
    void setPosition(const Vector3& p) { position_ = p; }
    int gpuOpcode() const override { return BIOIK_GOAL_POSITION; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {position_.x, position_.y, position_.z}); }
};

class OrientationGoal : public LinkGoalBase {  // :99-124
    Quaternion orientation_;

public:
    OrientationGoal() {}
    OrientationGoal(const std::string& link_name, const Quaternion& orientation, double weight = 1.0)
        : LinkGoalBase(link_name, weight), orientation_(orientation.normalized()) {}
    const Quaternion& getOrientation() const { return orientation_; }
    void setOrientation(const Quaternion& q) { orientation_ = q.normalized(); }
    int gpuOpcode() const override { return BIOIK_GOAL_ORIENTATION; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {orientation_.x, orientation_.y, orientation_.z, orientation_.w}); }
};

class PoseGoal : public LinkGoalBase {  // :126-181
    Vector3 position_;
    Quaternion orientation_;
    double rotation_scale_ = 0.5;

public:
    PoseGoal() {}
    PoseGoal(const std::string& link_name, const Vector3& position, const Quaternion& orientation, double weight = 1.0)
        : LinkGoalBase(link_name, weight), position_(position), orientation_(orientation.normalized()) {}
    const Vector3& getPosition() const { return position_; }
    void setPosition(const Vector3& p) { position_ = p; }
    const Quaternion& getOrientation() const { return orientation_; }
    void setOrientation(const Quaternion& q) { orientation_ = q.normalized(); }
    double getRotationScale() const { return rotation_scale_; }
    void setRotationScale(double s) { rotation_scale_ = s; }
    int gpuOpcode() const override { return BIOIK_GOAL_POSE; }
    void gpuParams(std::vector<double>& o) const override {
        o.insert(o.end(), {position_.x, position_.y, position_.z, orientation_.x, orientation_.y, orientation_.z, orientation_.w, rotation_scale_});
    }
};

class LookAtGoal : public LinkGoalBase {  // :183-212
    Vector3 axis_{1, 0, 0}, target_;

public:
    LookAtGoal() {}
    LookAtGoal(const std::string& link_name, const Vector3& axis, const Vector3& target, double weight = 1.0)
        : LinkGoalBase(link_name, weight), axis_(axis), target_(target) {}
    const Vector3& getAxis() const { return axis_; }
    const Vector3& getTarget() const { return target_; }
    void setAxis(const Vector3& a) { axis_ = a.normalized(); }
    void setTarget(const Vector3& t) { target_ = t; }
    int gpuOpcode() const override { return BIOIK_GOAL_LOOK_AT; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {axis_.x, axis_.y, axis_.z, target_.x, target_.y, target_.z}); }
};

class MaxDistanceGoal : public LinkGoalBase {  // :214-241
protected:
    Vector3 target;
    double distance = 1;

public:
    MaxDistanceGoal() {}
    MaxDistanceGoal(const std::string& link_name, const Vector3& target_, double distance_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), target(target_), distance(distance_) {}
    const Vector3& getTarget() const { return target; }
    void setTarget(const Vector3& t) { target = t; }
    double getDistance() const { return distance; }
    void setDistance(double d) { distance = d; }
    int gpuOpcode() const override { return BIOIK_GOAL_MAX_DISTANCE; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {target.x, target.y, target.z, distance}); }
};

class MinDistanceGoal : public MaxDistanceGoal {  // :243-270
public:
    using MaxDistanceGoal::MaxDistanceGoal;
    int gpuOpcode() const override { return BIOIK_GOAL_MIN_DISTANCE; }
};

class LineGoal : public LinkGoalBase {  // :272-298
    Vector3 position, direction{1, 0, 0};

public:
    LineGoal() {}
    LineGoal(const std::string& link_name, const Vector3& position_, const Vector3& direction_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), position(position_), direction(direction_.normalized()) {}
    const Vector3& getPosition() const { return position; }
    void setPosition(const Vector3& p) { position = p; }
    const Vector3& getDirection() const { return direction; }
    void setDirection(const Vector3& d) { direction = d.normalized(); }
    int gpuOpcode() const override { return BIOIK_GOAL_LINE; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {position.x, position.y, position.z, direction.x, direction.y, direction.z}); }
};

class PlaneGoal : public LinkGoalBase {  // :300-328
    Vector3 position, normal{0, 0, 1};

public:
    PlaneGoal() {}
    PlaneGoal(const std::string& link_name, const Vector3& position_, const Vector3& normal_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), position(position_), normal(normal_.normalized()) {}
    const Vector3& getPosition() const { return position; }
    void setPosition(const Vector3& p) { position = p; }
    const Vector3& getNormal() const { return normal; }
    void setNormal(const Vector3& n) { normal = n.normalized(); }
    int gpuOpcode() const override { return BIOIK_GOAL_PLANE; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {position.x, position.y, position.z, normal.x, normal.y, normal.z}); }
};

class AvoidJointLimitsGoal : public Goal {  // :379-402
public:
    AvoidJointLimitsGoal(double weight = 1.0, bool secondary = true) {
        weight_ = weight;
        secondary_ = secondary;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_AVOID_JOINT_LIMITS; }
};
class CenterJointsGoal : public Goal {  // :404-426
public:
    CenterJointsGoal(double weight = 1.0, bool secondary = true) {
        weight_ = weight;
        secondary_ = secondary;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_CENTER_JOINTS; }
};
class RegularizationGoal : public Goal {  // :428-445
public:
    RegularizationGoal(double weight = 1.0) { weight_ = weight; }
    int gpuOpcode() const override { return BIOIK_GOAL_REGULARIZATION; }
};
class MinimalDisplacementGoal : public Goal {  // :447-466
public:
    MinimalDisplacementGoal(double weight = 1.0, bool secondary = true) {
        weight_ = weight;
        secondary_ = secondary;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_MINIMAL_DISPLACEMENT; }
};

class JointVariableGoal : public Goal {  // :468-499
    std::string variable_name;
    double variable_position = 0;

public:
    JointVariableGoal() {}
    JointVariableGoal(const std::string& variable_name_, double variable_position_, double weight = 1.0, bool secondary = false)
        : variable_name(variable_name_), variable_position(variable_position_) {
        weight_ = weight;
        secondary_ = secondary;
    }
    double getVariablePosition() const { return variable_position; }
    void setVariablePosition(double p) { variable_position = p; }
    const std::string& getVariableName() const { return variable_name; }
    void setVariableName(const std::string& n) { variable_name = n; }
    int gpuOpcode() const override { return BIOIK_GOAL_JOINT_VARIABLE; }
    std::string gpuVariableName() const override { return variable_name; }
    void gpuParams(std::vector<double>& o) const override { o.push_back(variable_position); }
};

class SideGoal : public LinkGoalBase {  // :585-614 (constructors do not normalise, setters do)
protected:
    Vector3 axis{0, 0, 1}, direction{0, 0, 1};

public:
    SideGoal() {}
    SideGoal(const std::string& link_name, const Vector3& axis_, const Vector3& direction_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), axis(axis_), direction(direction_) {}
    const Vector3& getAxis() const { return axis; }
    const Vector3& getDirection() const { return direction; }
    void setAxis(const Vector3& a) { axis = a.normalized(); }
    void setDirection(const Vector3& d) { direction = d.normalized(); }
    int gpuOpcode() const override { return BIOIK_GOAL_SIDE; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {axis.x, axis.y, axis.z, direction.x, direction.y, direction.z}); }
};
class DirectionGoal : public SideGoal {  // :616-644
public:
    using SideGoal::SideGoal;
    int gpuOpcode() const override { return BIOIK_GOAL_DIRECTION; }
};

class ConeGoal : public LinkGoalBase {  // :646-712
    Vector3 position, axis{0, 0, 1}, direction{0, 0, 1};
    double position_weight = 0, angle = 0;

public:
    ConeGoal() {}
    ConeGoal(const std::string& link_name, const Vector3& axis_, const Vector3& direction_, double angle_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), axis(axis_), direction(direction_), angle(angle_) {}
    ConeGoal(const std::string& link_name, const Vector3& position_, const Vector3& axis_, const Vector3& direction_, double angle_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), position(position_), axis(axis_), direction(direction_), position_weight(1), angle(angle_) {}
    ConeGoal(const std::string& link_name, const Vector3& position_, double position_weight_, const Vector3& axis_, const Vector3& direction_, double angle_,
             double weight = 1.0)
        : LinkGoalBase(link_name, weight), position(position_), axis(axis_), direction(direction_), position_weight(position_weight_), angle(angle_) {}
    void setPosition(const Vector3& p) { position = p; }
    void setPositionWeight(double w) { position_weight = w; }
    void setAxis(const Vector3& a) { axis = a.normalized(); }
    void setDirection(const Vector3& d) { direction = d.normalized(); }
    void setAngle(double a) { angle = a; }
    int gpuOpcode() const override { return BIOIK_GOAL_CONE; }
    void gpuParams(std::vector<double>& o) const override {
        o.insert(o.end(), {position.x, position.y, position.z, position_weight, axis.x, axis.y, axis.z, direction.x, direction.y, direction.z, angle});
    }
};

// host-callback goals of the reference (TouchGoal :330-377, JointFunctionGoal :501-546, LinkFunctionGoal :548-583) keep
// gpuOpcode() == -1: the plugin refuses them with BIOIK_ERR_UNSUPPORTED.

class BalanceGoal : public Goal {  // goal_types.h:540-566, goal_types.cpp:231-272
    Vector3 target_{0, 0, 0}, axis_{0, 0, 1};

public:
    BalanceGoal() {}
    BalanceGoal(const Vector3& target, double weight = 1.0) : target_(target) { weight_ = weight; }
    const Vector3& getTarget() const { return target_; }
    const Vector3& getAxis() const { return axis_; }
    void setTarget(const Vector3& target) { target_ = target; }
    void setAxis(const Vector3& axis) { axis_ = axis; }
    int gpuOpcode() const override { return BIOIK_GOAL_BALANCE; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {target_.x, target_.y, target_.z, axis_.x, axis_.y, axis_.z}); }
};

}  // namespace bio_ik
