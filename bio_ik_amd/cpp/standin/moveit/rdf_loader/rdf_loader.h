// stand-in for moveit/rdf_loader/rdf_loader.h: the robot description and its semantic description from the parameter server (the
// stand-in ros::param_store(), keys <name> and <name>_semantic), and the RobotModel(urdf, srdf) constructor the plugin's string
// overload of initialize() uses with them (reference src/kinematics_plugin.cpp:167-189).  The stand-in model is built by the
// repository's own URDF / SRDF reader (bio_ik/urdf.h); see ../../README.md.
#pragma once
#include <bio_ik/urdf.h>
#include <moveit/robot_model/robot_model.h>
#include <ros/ros.h>

namespace rdf_loader {
class RDFLoader {
    urdf::ModelInterfaceSharedPtr urdf_;
    srdf::ModelSharedPtr srdf_;

public:
    explicit RDFLoader(const std::string& robot_description = "robot_description") {
        ros::NodeHandle nh;
        const std::string u = nh.param(robot_description, std::string()), s = nh.param(robot_description + "_semantic", std::string());
        if (!u.empty()) urdf_ = std::make_shared<urdf::ModelInterface>(), urdf_->xml_ = u;
        if (!s.empty()) srdf_ = std::make_shared<srdf::Model>(), srdf_->xml_ = s;
    }
    const urdf::ModelInterfaceSharedPtr& getURDF() const { return urdf_; }
    const srdf::ModelSharedPtr& getSRDF() const { return srdf_; }
};
}  // namespace rdf_loader

inline moveit::core::RobotModel::RobotModel(const urdf::ModelInterfaceSharedPtr& urdf_model, const srdf::ModelSharedPtr& srdf_model) {
    const std::shared_ptr<bio_ik::RobotModel> flat = bio_ik::loadURDF(urdf_model->xml_, srdf_model ? srdf_model->xml_ : std::string());
    urdf_->xml_ = urdf_model->xml_;
    for (size_t l = 0; l < flat->link_names.size(); l++) {
        const int t = flat->joint_type[l];
        const int fv = flat->joint_first_variable[l];
        std::string type = "fixed";
        if (t == BIOIK_JOINT_REVOLUTE) type = flat->var_bounded[fv] ? "revolute" : "continuous";
        else if (t == BIOIK_JOINT_PRISMATIC) type = "prismatic";
        else if (t != BIOIK_JOINT_FIXED) throw std::runtime_error("stand-in RobotModel: floating / planar joints are not modelled");
        const double* o = &flat->link_origin[l * 7];
        Eigen::Isometry3d origin;
        origin.linear() = Eigen::Quaterniond(o[6], o[3], o[4], o[5]).toRotationMatrix();
        origin.translation() = Eigen::Vector3d(o[0], o[1], o[2]);
        const std::string parent = flat->link_parent[l] >= 0 ? flat->link_names[flat->link_parent[l]] : std::string();
        addLink(flat->link_names[l], parent, flat->joint_names[l], type, origin, flat->joint_axis[l * 3], flat->joint_axis[l * 3 + 1], flat->joint_axis[l * 3 + 2],
                fv >= 0 ? flat->var_min[fv] : 0.0, fv >= 0 ? flat->var_max[fv] : 0.0, fv >= 0 ? flat->var_max_velocity[fv] : 0.0);
        if (!flat->link_mass.empty() && flat->link_mass[l] > 0)
            setInertial(flat->link_names[l], flat->link_mass[l], flat->link_center[l * 3], flat->link_center[l * 3 + 1], flat->link_center[l * 3 + 2]);
    }
    for (size_t l = 0; l < flat->link_names.size(); l++)
        if (flat->joint_mimic[l] >= 0) setMimic(flat->joint_names[l], flat->joint_names[flat->joint_mimic[l]], flat->joint_mimic_factor[l], flat->joint_mimic_offset[l]);
    for (auto& kv : flat->groups) {
        std::vector<std::string> joints, tips;
        for (int j : kv.second.active_joints) joints.push_back(flat->joint_names[j]);
        for (int tl : kv.second.tips) tips.push_back(flat->link_names[tl]);
        addJointsGroup(kv.first, joints, std::vector<std::string>());  // (tips of a group are not SRDF end effectors)
        (void)tips;
    }
}
