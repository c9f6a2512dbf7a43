// stand-in: RobotState is only named by RobotFK_MoveIt (a comparison class bio2 never instantiates)
#pragma once
#include <moveit/robot_model/robot_model.h>
namespace moveit {
namespace core {
class RobotState {
public:
    std::vector<double> positions_;
    Eigen::Isometry3d dummy_;
    explicit RobotState(const RobotModelConstPtr& m) : positions_(m ? m->getVariableCount() : 0) {}
    void setVariablePositions(const std::vector<double>& p) { positions_ = p; }
    const double* getVariablePositions() const { return positions_.data(); }
    void update() {}
    void setToDefaultValues() {}
    const Eigen::Isometry3d& getGlobalLinkTransform(const LinkModel*) const { throw std::runtime_error("RobotState stand-in has no FK"); }
    const Eigen::Isometry3d& getGlobalLinkTransform(const std::string&) const { throw std::runtime_error("RobotState stand-in has no FK"); }
};
typedef std::shared_ptr<RobotState> RobotStatePtr;
}  // namespace core
}  // namespace moveit
namespace robot_state = moveit::core;
