// bio_ik/robot_model.h — flat robot description for the C-ABI (stand-in for what the reference reads from
// moveit::core::RobotModel / JointModelGroup: src/forward_kinematics.h:192-213, include/bio_ik/robot_info.h:70-106,
// src/problem.cpp:117-124,201-204).  With MoveIt available the same arrays are filled from a RobotModel.
#pragma once
#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include <bioik_hip.h>

namespace bio_ik {

struct JointModelGroup {
    std::string name;
    std::vector<int32_t> active_joints;  // link indices, getActiveJointModels() order
    std::vector<int32_t> tips;
};

class RobotModel {
public:
    std::vector<std::string> link_names, joint_names, variable_names;
    std::vector<int32_t> link_parent, joint_type, joint_first_variable, joint_mimic;
    std::vector<double> link_origin, joint_axis, joint_mimic_factor, joint_mimic_offset, var_min, var_max, var_max_velocity;
    std::vector<double> link_mass, link_center;  // urdf <inertial> mass / origin xyz per link (BalanceGoal); empty = none
    std::vector<uint8_t> var_bounded;
    std::map<std::string, JointModelGroup> groups;

    static void quatFromRpy(double r, double p, double y, double* q) {
        double cr = std::cos(r / 2), sr = std::sin(r / 2), cp = std::cos(p / 2), sp = std::sin(p / 2), cy = std::cos(y / 2), sy = std::sin(y / 2);
        q[0] = sr * cp * cy - cr * sp * sy, q[1] = cr * sp * cy + sr * cp * sy, q[2] = cr * cp * sy - sr * sp * cy, q[3] = cr * cp * cy + sr * sp * sy;
    }
    int linkIndex(const std::string& n) const {
        for (size_t i = 0; i < link_names.size(); i++)
            if (link_names[i] == n) return (int)i;
        throw std::runtime_error("link not found: " + n);  // reference problem.cpp:141
    }
    int jointIndex(const std::string& n) const {
        for (size_t i = 0; i < joint_names.size(); i++)
            if (joint_names[i] == n) return (int)i;
        throw std::runtime_error("joint not found: " + n);
    }
    int variableIndex(const std::string& n) const {
        for (size_t i = 0; i < variable_names.size(); i++)
            if (variable_names[i] == n) return (int)i;
        throw std::runtime_error("joint variable not found: " + n);  // reference problem.cpp:125
    }
    // type: "fixed" | "revolute" | "continuous" | "prismatic" (URDF semantics) | "floating" | "planar" (MoveIt's multi-variable joints:
    // variables <joint>/trans_x .. rot_w and <joint>/x, y, theta; the device takes one such joint, at the root of the model)
    int addLink(const std::string& link, const std::string& parent, const std::string& joint, const std::string& type, const double (&xyz)[3],
                const double (&rpy)[3], const double (&axis)[3], double lower = 0, double upper = 0, double velocity = 0) {
        int idx = (int)link_names.size();
        link_names.push_back(link);
        joint_names.push_back(joint.empty() ? link + "_joint" : joint);
        link_parent.push_back(parent.empty() ? -1 : linkIndex(parent));
        double q[4];
        quatFromRpy(rpy[0], rpy[1], rpy[2], q);
        for (double v : {xyz[0], xyz[1], xyz[2], q[0], q[1], q[2], q[3]}) link_origin.push_back(v);
        int t = type == "fixed" ? BIOIK_JOINT_FIXED : (type == "prismatic" ? BIOIK_JOINT_PRISMATIC : (type == "floating" ? BIOIK_JOINT_FLOATING : (type == "planar" ? BIOIK_JOINT_PLANAR : BIOIK_JOINT_REVOLUTE)));
        if (type != "fixed" && type != "prismatic" && type != "revolute" && type != "continuous" && type != "floating" && type != "planar")
            throw std::runtime_error("unsupported joint type " + type);
        joint_type.push_back(t);
        double n = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
        const bool has_axis = t == BIOIK_JOINT_REVOLUTE || t == BIOIK_JOINT_PRISMATIC;
        for (int c = 0; c < 3; c++) joint_axis.push_back(!has_axis || n == 0 ? axis[c] : axis[c] / n);
        joint_mimic.push_back(-1);
        joint_mimic_factor.push_back(1.0);
        joint_mimic_offset.push_back(0.0);
        auto variable = [&](const std::string& name, double lo, double hi, bool bounded) {
            variable_names.push_back(name), var_min.push_back(lo), var_max.push_back(hi), var_bounded.push_back(bounded ? 1 : 0), var_max_velocity.push_back(velocity);
        };
        if (t == BIOIK_JOINT_FIXED) {
            joint_first_variable.push_back(-1);
        } else if (t == BIOIK_JOINT_FLOATING) {  // MoveIt FloatingJointModel: position unbounded, quaternion components in [-1, 1]
            joint_first_variable.push_back((int)variable_names.size());
            for (const char* c : {"trans_x", "trans_y", "trans_z"}) variable(joint_names.back() + "/" + c, -1e300, 1e300, false);
            for (const char* c : {"rot_x", "rot_y", "rot_z", "rot_w"}) variable(joint_names.back() + "/" + c, -1.0, 1.0, true);
        } else if (t == BIOIK_JOINT_PLANAR) {  // PlanarJointModel: x, y unbounded, theta in [-pi, pi] and not position-bounded
            joint_first_variable.push_back((int)variable_names.size());
            variable(joint_names.back() + "/x", -1e300, 1e300, false), variable(joint_names.back() + "/y", -1e300, 1e300, false);
            variable(joint_names.back() + "/theta", -M_PI, M_PI, false);
        } else {
            joint_first_variable.push_back((int)variable_names.size());
            variable_names.push_back(joint_names.back());
            bool cont = type == "continuous";  // MoveIt: bounds [-pi, pi], position_bounded_ = false
            var_min.push_back(cont ? -M_PI : lower);
            var_max.push_back(cont ? M_PI : upper);
            var_bounded.push_back(cont ? 0 : 1);
            var_max_velocity.push_back(velocity);
        }
        return idx;
    }
    void setInertial(const std::string& link, double mass, double cx, double cy, double cz) {  // urdf <inertial>
        link_mass.resize(link_names.size(), 0.0), link_center.resize(3 * link_names.size(), 0.0);
        const int i = linkIndex(link);
        link_mass[i] = mass, link_center[3 * i] = cx, link_center[3 * i + 1] = cy, link_center[3 * i + 2] = cz;
    }
    void addChainGroup(const std::string& name, const std::string& base, const std::string& tip) {
        JointModelGroup g;
        g.name = name;
        std::vector<int32_t> links;
        for (int l = linkIndex(tip), b = linkIndex(base); l != b; l = link_parent[l]) {
            if (l < 0) throw std::runtime_error(base + " is not an ancestor of " + tip);
            links.insert(links.begin(), l);
        }
        for (int l : links)  // JointModelGroup::getActiveJointModels: neither fixed nor mimic joints
            if (joint_type[l] != BIOIK_JOINT_FIXED && joint_mimic[l] < 0) g.active_joints.push_back(l);
        g.tips.push_back(linkIndex(tip));
        groups[name] = g;
    }
    void addJointGroup(const std::string& name, const std::vector<std::string>& joints, const std::vector<std::string>& tips) {
        JointModelGroup g;
        g.name = name;
        for (auto& j : joints) g.active_joints.push_back(jointIndex(j));
        for (auto& t : tips) g.tips.push_back(linkIndex(t));
        groups[name] = g;
    }
    std::vector<double> defaultPositions() const {  // RobotModel::getVariableDefaultPositions
        std::vector<double> out(variable_names.size(), 0.0);
        for (size_t v = 0; v < out.size(); v++)
            if (!(var_min[v] <= 0.0 && 0.0 <= var_max[v])) out[v] = 0.5 * (var_min[v] + var_max[v]);
        for (size_t l = 0; l < joint_type.size(); l++)
            if (joint_type[l] == BIOIK_JOINT_FLOATING) out[joint_first_variable[l] + 6] = 1.0;  // identity quaternion
        return out;
    }
    // A joint that mimics a joint that itself mimics another follows the joint at the end of the chain with the composed factor and offset, as MoveIt's
    // RobotModel::buildMimic leaves every model: x = f1 (f2 y + o2) + o1 -> factor f1 f2, offset o1 + f1 o2 (the loaders call it once the joints exist)
    void resolveMimicChains() {
        for (size_t pass = 0; pass <= joint_mimic.size() + 1; pass++) {
            bool changed = false;
            for (size_t i = 0; i < joint_mimic.size(); i++) {
                const int32_t m = joint_mimic[i];
                if (m >= 0 && joint_mimic[m] >= 0) {
                    joint_mimic_offset[i] = joint_mimic_offset[i] + joint_mimic_factor[i] * joint_mimic_offset[m];
                    joint_mimic_factor[i] = joint_mimic_factor[i] * joint_mimic_factor[m];
                    joint_mimic[i] = joint_mimic[m];
                    changed = true;
                }
            }
            if (!changed) return;
        }
        throw std::runtime_error("mimic joints that follow each other in a circle");
    }
    bioik_model_desc desc() const {
        bioik_model_desc d{};
        d.struct_size = sizeof(d);
        d.n_links = (uint32_t)link_names.size();
        d.n_variables = (uint32_t)variable_names.size();
        d.link_parent = link_parent.data(), d.link_origin = link_origin.data(), d.joint_type = joint_type.data(), d.joint_axis = joint_axis.data();
        d.joint_first_variable = joint_first_variable.data(), d.joint_mimic = joint_mimic.data();
        d.joint_mimic_factor = joint_mimic_factor.data(), d.joint_mimic_offset = joint_mimic_offset.data();
        d.var_min = var_min.data(), d.var_max = var_max.data(), d.var_bounded = var_bounded.data(), d.var_max_velocity = var_max_velocity.data();
        if (link_mass.size() == link_names.size() && link_center.size() == 3 * link_names.size()) d.link_mass = link_mass.data(), d.link_center = link_center.data();
        return d;
    }
    // frame algebra for the plugin boundary (goal poses into the model frame, kinematics_plugin.cpp:487-502)
    static void rotate(const double* q, const double* v, double* r) {
        double tx = 2 * (q[1] * v[2] - q[2] * v[1]), ty = 2 * (q[2] * v[0] - q[0] * v[2]), tz = 2 * (q[0] * v[1] - q[1] * v[0]);
        r[0] = v[0] + q[3] * tx + q[1] * tz - q[2] * ty;
        r[1] = v[1] + q[3] * ty + q[2] * tx - q[0] * tz;
        r[2] = v[2] + q[3] * tz + q[0] * ty - q[1] * tx;
    }
    static void concat(const double* a, const double* b, double* r) {
        double p[3];
        rotate(a + 3, b, p);
        double o[7] = {a[0] + p[0], a[1] + p[1], a[2] + p[2],
                       a[6] * b[3] + a[3] * b[6] + a[4] * b[5] - a[5] * b[4], a[6] * b[4] - a[3] * b[5] + a[4] * b[6] + a[5] * b[3],
                       a[6] * b[5] + a[3] * b[4] - a[4] * b[3] + a[5] * b[6], a[6] * b[6] - a[3] * b[3] - a[4] * b[4] - a[5] * b[5]};
        for (int i = 0; i < 7; i++) r[i] = o[i];
    }
    void linkTransform(int link, const std::vector<double>& positions, double* f) const {
        std::vector<int> chain;
        for (int l = link; l >= 0; l = link_parent[l]) chain.insert(chain.begin(), l);
        double cur[7] = {0, 0, 0, 0, 0, 0, 1};
        for (int l : chain) {
            concat(cur, &link_origin[7 * l], cur);
            if (joint_type[l] == BIOIK_JOINT_REVOLUTE) {
                double h = 0.5 * positions[joint_first_variable[l]], s = std::sin(h);
                double j[7] = {0, 0, 0, joint_axis[3 * l] * s, joint_axis[3 * l + 1] * s, joint_axis[3 * l + 2] * s, std::cos(h)};
                concat(cur, j, cur);
            } else if (joint_type[l] == BIOIK_JOINT_PRISMATIC) {
                double v = positions[joint_first_variable[l]];
                double j[7] = {joint_axis[3 * l] * v, joint_axis[3 * l + 1] * v, joint_axis[3 * l + 2] * v, 0, 0, 0, 1};
                concat(cur, j, cur);
            } else if (joint_type[l] == BIOIK_JOINT_FLOATING) {
                const double* v = &positions[joint_first_variable[l]];
                const double n = std::sqrt(v[3] * v[3] + v[4] * v[4] + v[5] * v[5] + v[6] * v[6]);
                double j[7] = {v[0], v[1], v[2], v[3] / n, v[4] / n, v[5] / n, v[6] / n};
                concat(cur, j, cur);
            } else if (joint_type[l] == BIOIK_JOINT_PLANAR) {
                const double* v = &positions[joint_first_variable[l]];
                double j[7] = {v[0], v[1], 0, 0, 0, std::sin(0.5 * v[2]), std::cos(0.5 * v[2])};
                concat(cur, j, cur);
            }
        }
        for (int i = 0; i < 7; i++) f[i] = cur[i];
    }
};

}  // namespace bio_ik
