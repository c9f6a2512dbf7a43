// bio_ik/goal_types.h — the reference's built-in goal classes (include/bio_ik/goal_types.h:56-712) with the same names,
// constructors, setters, normalisation behaviour, `describe()` and `evaluate()`.  `evaluate()` is the closed form of the cost as the
// reference defines it (the solver kernels evaluate the same expression for a whole population, bio_ik_amd/csrc/bioik_device.h:
// goal_eval; tests/cpp hold one against the other); the gpu*() members serialise the goal for the device (bio_ik/goal.h).
#pragma once
#include <functional>

#include "goal.h"

namespace bio_ik {

class LinkGoalBase : public Goal {  // goal_types.h:56-78
    std::string link_name_;

public:
    LinkGoalBase() { weight_ = 1; }
    LinkGoalBase(const std::string& link_name, double weight) : link_name_(link_name) { weight_ = weight; }
    void setLinkName(const std::string& n) { link_name_ = n; }
    const std::string& getLinkName() const { return link_name_; }
    void describe(GoalContext& context) const override {
        Goal::describe(context);
        context.addLink(link_name_);
    }
    std::string gpuLinkName() const override { return link_name_; }
};

class PositionGoal : public LinkGoalBase {  // :80-97
    Vector3 position_;

public:
    PositionGoal() : position_(0, 0, 0) {}
    PositionGoal(const std::string& link_name, const Vector3& position, double weight = 1.0) : LinkGoalBase(link_name, weight), position_(position) {}
    const Vector3& getPosition() const { return position_; }
    void setPosition(const Vector3& p) { position_ = p; }
    double evaluate(const GoalContext& context) const override { return context.getLinkFrame().getPosition().distance2(getPosition()); }
    int gpuOpcode() const override { return BIOIK_GOAL_POSITION; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {position_.x(), position_.y(), position_.z()}); }
};

// the orientation cost of the reference: squared distance of the two quaternions, q and -q being the same rotation (:115-124)
inline double quaternionDistance2(const Quaternion& goal, const Quaternion& q) { return std::fmin((goal - q).length2(), (goal + q).length2()); }

class OrientationGoal : public LinkGoalBase {  // :99-124
    Quaternion orientation_;

public:
    OrientationGoal() : orientation_(0, 0, 0, 1) {}
    OrientationGoal(const std::string& link_name, const Quaternion& orientation, double weight = 1.0)
        : LinkGoalBase(link_name, weight), orientation_(orientation.normalized()) {}
    const Quaternion& getOrientation() const { return orientation_; }
    void setOrientation(const Quaternion& q) { orientation_ = q.normalized(); }
    double evaluate(const GoalContext& context) const override { return quaternionDistance2(getOrientation(), context.getLinkFrame().getOrientation()); }
    int gpuOpcode() const override { return BIOIK_GOAL_ORIENTATION; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {orientation_.x(), orientation_.y(), orientation_.z(), orientation_.w()}); }
};

class PoseGoal : public LinkGoalBase {  // :126-181
    Vector3 position_;
    Quaternion orientation_;
    double rotation_scale_ = 0.5;

public:
    PoseGoal() : position_(0, 0, 0), orientation_(0, 0, 0, 1) {}
    PoseGoal(const std::string& link_name, const Vector3& position, const Quaternion& orientation, double weight = 1.0)
        : LinkGoalBase(link_name, weight), position_(position), orientation_(orientation.normalized()) {}
    const Vector3& getPosition() const { return position_; }
    void setPosition(const Vector3& p) { position_ = p; }
    const Quaternion& getOrientation() const { return orientation_; }
    void setOrientation(const Quaternion& q) { orientation_ = q.normalized(); }
    double getRotationScale() const { return rotation_scale_; }
    void setRotationScale(double s) { rotation_scale_ = s; }
    double evaluate(const GoalContext& context) const override {
        const Frame& f = context.getLinkFrame();
        return f.getPosition().distance2(getPosition()) + quaternionDistance2(getOrientation(), f.getOrientation()) * (rotation_scale_ * rotation_scale_);
    }
    int gpuOpcode() const override { return BIOIK_GOAL_POSE; }
    void gpuParams(std::vector<double>& o) const override {
        o.insert(o.end(), {position_.x(), position_.y(), position_.z(), orientation_.x(), orientation_.y(), orientation_.z(), orientation_.w(), rotation_scale_});
    }
};

class LookAtGoal : public LinkGoalBase {  // :183-212
    Vector3 axis_, target_;

public:
    LookAtGoal() : axis_(1, 0, 0), target_(0, 0, 0) {}
    LookAtGoal(const std::string& link_name, const Vector3& axis, const Vector3& target, double weight = 1.0)
        : LinkGoalBase(link_name, weight), axis_(axis), target_(target) {}
    const Vector3& getAxis() const { return axis_; }
    const Vector3& getTarget() const { return target_; }
    void setAxis(const Vector3& a) { axis_ = a.normalized(); }
    void setTarget(const Vector3& t) { target_ = t; }
    double evaluate(const GoalContext& context) const override {
        const Frame& f = context.getLinkFrame();
        const Vector3 axis = quatRotate(f.getOrientation(), axis_);
        return (target_ - f.getPosition()).normalized().distance2(axis.normalized());
    }
    int gpuOpcode() const override { return BIOIK_GOAL_LOOK_AT; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {axis_.x(), axis_.y(), axis_.z(), target_.x(), target_.y(), target_.z()}); }
};

class MaxDistanceGoal : public LinkGoalBase {  // :214-241
protected:
    Vector3 target;
    double distance = 1;

public:
    MaxDistanceGoal() : target(0, 0, 0) {}
    MaxDistanceGoal(const std::string& link_name, const Vector3& target_, double distance_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), target(target_), distance(distance_) {}
    const Vector3& getTarget() const { return target; }
    void setTarget(const Vector3& t) { target = t; }
    double getDistance() const { return distance; }
    void setDistance(double d) { distance = d; }
    double evaluate(const GoalContext& context) const override {
        const double d = std::fmax(0.0, context.getLinkFrame().getPosition().distance(target) - distance);
        return d * d;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_MAX_DISTANCE; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {target.x(), target.y(), target.z(), distance}); }
};

class MinDistanceGoal : public MaxDistanceGoal {  // :243-270
public:
    using MaxDistanceGoal::MaxDistanceGoal;
    double evaluate(const GoalContext& context) const override {
        const double d = std::fmax(0.0, distance - context.getLinkFrame().getPosition().distance(target));
        return d * d;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_MIN_DISTANCE; }
};

class LineGoal : public LinkGoalBase {  // :272-298
    Vector3 position, direction;

public:
    LineGoal() : position(0, 0, 0), direction(1, 0, 0) {}
    LineGoal(const std::string& link_name, const Vector3& position_, const Vector3& direction_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), position(position_), direction(direction_.normalized()) {}
    const Vector3& getPosition() const { return position; }
    void setPosition(const Vector3& p) { position = p; }
    const Vector3& getDirection() const { return direction; }
    void setDirection(const Vector3& d) { direction = d.normalized(); }
    double evaluate(const GoalContext& context) const override {
        const Vector3& p = context.getLinkFrame().getPosition();
        return position.distance2(p - direction * direction.dot(p - position));
    }
    int gpuOpcode() const override { return BIOIK_GOAL_LINE; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {position.x(), position.y(), position.z(), direction.x(), direction.y(), direction.z()}); }
};

class PlaneGoal : public LinkGoalBase {  // :300-328
    Vector3 position, normal;

public:
    PlaneGoal() : position(0, 0, 0), normal(0, 0, 1) {}
    PlaneGoal(const std::string& link_name, const Vector3& position_, const Vector3& normal_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), position(position_), normal(normal_.normalized()) {}
    const Vector3& getPosition() const { return position; }
    void setPosition(const Vector3& p) { position = p; }
    const Vector3& getNormal() const { return normal; }
    void setNormal(const Vector3& n) { normal = n.normalized(); }
    double evaluate(const GoalContext& context) const override {
        const double d = (context.getLinkFrame().getPosition() - position).dot(normal);
        return d * d;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_PLANE; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {position.x(), position.y(), position.z(), normal.x(), normal.y(), normal.z()}); }
};

// ---- goals over the joint values (:379-499) -------------------------------------------------------------------------------------
class AvoidJointLimitsGoal : public Goal {  // :379-402
public:
    AvoidJointLimitsGoal(double weight = 1.0, bool secondary = true) {
        weight_ = weight;
        secondary_ = secondary;
    }
    double evaluate(const GoalContext& context) const override {
        const RobotInfo& info = context.getRobotInfo();
        double sum = 0.0;
        for (size_t i = 0; i < context.getProblemVariableCount(); i++) {
            const size_t ivar = context.getProblemVariableIndex(i);
            if (info.getClipMax(ivar) == DBL_MAX) continue;
            double d = context.getProblemVariablePosition(i) - (info.getMin(ivar) + info.getMax(ivar)) * 0.5;
            d = std::fmax(0.0, std::fabs(d) * 2.0 - info.getSpan(ivar) * 0.5);
            d *= context.getProblemVariableWeight(i);
            sum += d * d;
        }
        return sum;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_AVOID_JOINT_LIMITS; }
};
class CenterJointsGoal : public Goal {  // :404-426
public:
    CenterJointsGoal(double weight = 1.0, bool secondary = true) {
        weight_ = weight;
        secondary_ = secondary;
    }
    double evaluate(const GoalContext& context) const override {
        const RobotInfo& info = context.getRobotInfo();
        double sum = 0.0;
        for (size_t i = 0; i < context.getProblemVariableCount(); i++) {
            const size_t ivar = context.getProblemVariableIndex(i);
            if (info.getClipMax(ivar) == DBL_MAX) continue;
            double d = context.getProblemVariablePosition(i) - (info.getMin(ivar) + info.getMax(ivar)) * 0.5;
            d *= context.getProblemVariableWeight(i);
            sum += d * d;
        }
        return sum;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_CENTER_JOINTS; }
};
class RegularizationGoal : public Goal {  // :428-445
public:
    RegularizationGoal(double weight = 1.0) { weight_ = weight; }
    double evaluate(const GoalContext& context) const override {
        double sum = 0.0;
        for (size_t i = 0; i < context.getProblemVariableCount(); i++) {
            const double d = context.getProblemVariablePosition(i) - context.getProblemVariableInitialGuess(i);
            sum += d * d;
        }
        return sum;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_REGULARIZATION; }
};
class MinimalDisplacementGoal : public Goal {  // :447-466
public:
    MinimalDisplacementGoal(double weight = 1.0, bool secondary = true) {
        weight_ = weight;
        secondary_ = secondary;
    }
    double evaluate(const GoalContext& context) const override {
        double sum = 0.0;
        for (size_t i = 0; i < context.getProblemVariableCount(); i++) {
            double d = context.getProblemVariablePosition(i) - context.getProblemVariableInitialGuess(i);
            d *= context.getProblemVariableWeight(i);
            sum += d * d;
        }
        return sum;
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
    void describe(GoalContext& context) const override {
        Goal::describe(context);
        context.addVariable(variable_name);
    }
    double evaluate(const GoalContext& context) const override {
        const double d = variable_position - context.getVariablePosition();
        return d * d;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_JOINT_VARIABLE; }
    std::string gpuVariableName() const override { return variable_name; }
    void gpuParams(std::vector<double>& o) const override { o.push_back(variable_position); }
};

// ---- host-callback goals (:501-583): any function of the joint values / of a link frame.  No device opcode: they are evaluated on
//      the host (bio_ik/goal_eval.h) and cannot be part of a device solve (the plugin refuses them with a message). ----------------
class JointFunctionGoal : public Goal {  // :501-538
    std::vector<std::string> variable_names;
    std::function<void(std::vector<double>&)> function;

public:
    JointFunctionGoal() {}
    JointFunctionGoal(const std::vector<std::string>& variable_names_, const std::function<void(std::vector<double>&)>& function_, double weight = 1.0,
                      bool secondary = false)
        : variable_names(variable_names_), function(function_) {
        weight_ = weight;
        secondary_ = secondary;
    }
    void setJointVariableNames(const std::vector<std::string>& n) { variable_names = n; }
    void setJointVariableFunction(const std::function<void(std::vector<double>&)>& f) { function = f; }
    void describe(GoalContext& context) const override {
        Goal::describe(context);
        for (auto& name : variable_names) context.addVariable(name);
    }
    double evaluate(const GoalContext& context) const override {  // squared distance between the values and what the function makes of them
        std::vector<double>& temp = context.getTempVector();
        temp.resize(variable_names.size());
        for (size_t i = 0; i < variable_names.size(); i++) temp[i] = context.getVariablePosition(i);
        function(temp);
        double sum = 0.0;
        for (size_t i = 0; i < variable_names.size(); i++) {
            const double d = temp[i] - context.getVariablePosition(i);
            sum += d * d;
        }
        return sum;
    }
};
class LinkFunctionGoal : public LinkGoalBase {  // :570-583
    std::function<double(const Vector3&, const Quaternion&)> function;

public:
    LinkFunctionGoal() {}
    LinkFunctionGoal(const std::string& link_name, const std::function<double(const Vector3&, const Quaternion&)>& function_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), function(function_) {}
    void setLinkFunction(const std::function<double(const Vector3&, const Quaternion&)>& f) { function = f; }
    double evaluate(const GoalContext& context) const override { return function(context.getLinkFrame().getPosition(), context.getLinkFrame().getOrientation()); }
};

// ---- direction goals (:585-712) -------------------------------------------------------------------------------------------------
class SideGoal : public LinkGoalBase {  // :585-614 (constructors do not normalise, setters do)
protected:
    Vector3 axis, direction;

public:
    SideGoal() : axis(0, 0, 1), direction(0, 0, 1) {}
    SideGoal(const std::string& link_name, const Vector3& axis_, const Vector3& direction_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), axis(axis_), direction(direction_) {}
    const Vector3& getAxis() const { return axis; }
    const Vector3& getDirection() const { return direction; }
    void setAxis(const Vector3& a) { axis = a.normalized(); }
    void setDirection(const Vector3& d) { direction = d.normalized(); }
    double evaluate(const GoalContext& context) const override {
        const double f = std::fmax(0.0, quatRotate(context.getLinkFrame().getOrientation(), axis).dot(direction));
        return f * f;
    }
    int gpuOpcode() const override { return BIOIK_GOAL_SIDE; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {axis.x(), axis.y(), axis.z(), direction.x(), direction.y(), direction.z()}); }
};
class DirectionGoal : public SideGoal {  // :616-644
public:
    using SideGoal::SideGoal;
    double evaluate(const GoalContext& context) const override { return quatRotate(context.getLinkFrame().getOrientation(), axis).distance2(direction); }
    int gpuOpcode() const override { return BIOIK_GOAL_DIRECTION; }
};

class ConeGoal : public LinkGoalBase {  // :646-712
    Vector3 position, axis, direction;
    double position_weight = 0, angle = 0;

public:
    ConeGoal() : position(0, 0, 0), axis(0, 0, 1), direction(0, 0, 1) {}
    ConeGoal(const std::string& link_name, const Vector3& axis_, const Vector3& direction_, double angle_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), position(0, 0, 0), axis(axis_), direction(direction_), angle(angle_) {}
    ConeGoal(const std::string& link_name, const Vector3& position_, const Vector3& axis_, const Vector3& direction_, double angle_, double weight = 1.0)
        : LinkGoalBase(link_name, weight), position(position_), axis(axis_), direction(direction_), position_weight(1), angle(angle_) {}
    ConeGoal(const std::string& link_name, const Vector3& position_, double position_weight_, const Vector3& axis_, const Vector3& direction_, double angle_,
             double weight = 1.0)
        : LinkGoalBase(link_name, weight), position(position_), axis(axis_), direction(direction_), position_weight(position_weight_), angle(angle_) {}
    const Vector3& getPosition() const { return position; }
    double getPositionWeight() const { return position_weight; }
    const Vector3& getAxis() const { return axis; }
    const Vector3& getDirection() const { return direction; }
    double getAngle() const { return angle; }
    void setPosition(const Vector3& p) { position = p; }
    void setPositionWeight(double w) { position_weight = w; }
    void setAxis(const Vector3& a) { axis = a.normalized(); }
    void setDirection(const Vector3& d) { direction = d.normalized(); }
    void setAngle(double a) { angle = a; }
    double evaluate(const GoalContext& context) const override {
        const Frame& f = context.getLinkFrame();
        const Vector3 v = quatRotate(f.getOrientation(), axis);
        const double c = v.dot(direction) / std::sqrt(v.length2() * direction.length2());
        const double d = std::fmax(0.0, std::acos(std::fmin(1.0, std::fmax(-1.0, c))) - angle);
        return d * d + position_weight * position_weight * (position - f.getPosition()).length2();
    }
    int gpuOpcode() const override { return BIOIK_GOAL_CONE; }
    void gpuParams(std::vector<double>& o) const override {
        o.insert(o.end(), {position.x(), position.y(), position.z(), position_weight, axis.x(), axis.y(), axis.z(), direction.x(), direction.y(), direction.z(), angle});
    }
};

// TouchGoal (:330-377) needs FCL collision shapes of the robot's links: not provided by this build.

class BalanceGoal : public Goal {  // goal_types.h:540-566, goal_types.cpp:231-272
    Vector3 target_, axis_;

public:
    BalanceGoal() : target_(0, 0, 0), axis_(0, 0, 1) {}
    BalanceGoal(const Vector3& target, double weight = 1.0) : target_(target), axis_(0, 0, 1) { weight_ = weight; }
    const Vector3& getTarget() const { return target_; }
    const Vector3& getAxis() const { return axis_; }
    void setTarget(const Vector3& target) { target_ = target; }
    void setAxis(const Vector3& axis) { axis_ = axis; }
    // (its cost reads every link with a mass: evaluated on the device only, where the problem compiler knows the model's inertials;
    // the host-side evaluate() of this build returns 0 for it)
    int gpuOpcode() const override { return BIOIK_GOAL_BALANCE; }
    void gpuParams(std::vector<double>& o) const override { o.insert(o.end(), {target_.x(), target_.y(), target_.z(), axis_.x(), axis_.y(), axis_.z()}); }
};

}  // namespace bio_ik
