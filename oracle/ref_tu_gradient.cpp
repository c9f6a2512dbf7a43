// ORACLE SUPPORT: compiles the reference's src/ik_gradient.cpp, unmodified, from where it lies (IKGradientDescent: gd / gd_c / gd_r,
// IKJacobian: jac; the least-squares step of `jac` goes through the stand-in JacobiSVD of ref_shim/Eigen/Dense)
#include "ref_prelude.h"
#include "ik_gradient.cpp"
