// stand-in for moveit_msgs/MoveItErrorCodes.h (see ../README.md)
#pragma once
namespace moveit_msgs {
struct MoveItErrorCodes {
    enum { SUCCESS = 1, FAILURE = 99999, TIMED_OUT = -6, NO_IK_SOLUTION = -31 };
    int val = 0;
};
}  // namespace moveit_msgs
