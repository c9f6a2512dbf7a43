// ref_prelude.h — ORACLE SUPPORT.  Included first by every translation unit of oracle/_ref.
//
// The reference selects hand-written AVX+FMA / SSE2 variants of its linearised-FK kernels through GCC function
// multiversioning when it sees an x86 target (src/forward_kinematics.h:43-59).  g++ 11 cannot assemble those in-class
// multiversioned definitions ("symbol is already defined"), and the FMA variant would round differently from the
// scalar one anyway.  All standard / stand-in headers are therefore pulled in here, and the x86 target macros are then
// hidden, so that the reference's own preprocessor logic picks its portable scalar variants (`FUNCTION_MULTIVERSIONING 0`)
// — the reference sources themselves stay untouched.
#pragma once
#include <malloc.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <csignal>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <typeindex>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <Eigen/Dense>
#include <XmlRpcException.h>
#include <geometric_shapes/bodies.h>
#include <geometric_shapes/shapes.h>
#include <kdl/treefksolverpos_recursive.hpp>
#include <kdl_parser/kdl_parser.hpp>
#include <moveit/collision_detection/collision_common.h>
#include <moveit/collision_detection_fcl/collision_common.h>
#include <moveit/kinematics_base/kinematics_base.h>
#include <moveit/robot_model/joint_model_group.h>
#include <moveit/robot_model/robot_model.h>
#include <moveit/robot_state/robot_state.h>
#include <ros/ros.h>
#include <tf2/LinearMath/Quaternion.h>
#include <tf2/LinearMath/Vector3.h>
#include <tf2_geometry_msgs/tf2_geometry_msgs.h>
#include <tf2_kdl/tf2_kdl.h>
#include <tf_conversions/tf_kdl.h>

#if !defined(REF_NATIVE_KERNELS)  // the timing build of the solver keeps the reference's own AVX/SSE2 dispatch
#undef __x86_64__
#undef __i386__
#endif
