// stand-in for moveit/kinematics_base/kinematics_base.h: the option struct goal.h derives from
#pragma once
#include <moveit/robot_model/robot_model.h>
#include <moveit/robot_state/robot_state.h>
#include <tf2_geometry_msgs/tf2_geometry_msgs.h>
namespace moveit_msgs {
struct MoveItErrorCodes {
    enum { SUCCESS = 1, NO_IK_SOLUTION = -31 };
    int val = 0;
};
}  // namespace moveit_msgs
namespace kinematics {
struct KinematicsQueryOptions {
    bool lock_redundant_joints = false;
    bool return_approximate_solution = false;
};
}  // namespace kinematics
