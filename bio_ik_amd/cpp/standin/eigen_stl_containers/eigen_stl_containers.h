// stand-in for eigen_stl_containers: the aligned-vector typedefs MoveIt's RobotState::setFromIK takes (see ../README.md)
#pragma once
#include <vector>

#include <Eigen/Geometry>
namespace EigenSTL {
typedef std::vector<Eigen::Affine3d> vector_Affine3d;
typedef std::vector<Eigen::Isometry3d> vector_Isometry3d;
}  // namespace EigenSTL
