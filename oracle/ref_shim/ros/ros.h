// stand-in for ros/ros.h: wall clock only
#pragma once
#include <algorithm>
#include <chrono>
#include <functional>
#include <iostream>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <moveit/robot_model/robot_model.h>
namespace ros {
struct WallTime {
    double t = 0;
    static WallTime now() {
        WallTime w;
        w.t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        return w;
    }
    double toSec() const { return t; }
};
}  // namespace ros
