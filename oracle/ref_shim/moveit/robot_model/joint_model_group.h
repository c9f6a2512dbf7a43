#pragma once
#include <moveit/robot_model/robot_model.h>
