// stand-in: JointModelGroup lives in the robot_model stand-in
#pragma once
#include <moveit/robot_model/robot_model.h>
