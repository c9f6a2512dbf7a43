#pragma once
#include <moveit/collision_detection/collision_common.h>
