// orc_model.h — ORACLE (test infrastructure): robot model, RobotInfo and the L2 kinematics engine.
//
// Restates reference include/bio_ik/robot_info.h:70-113 and src/forward_kinematics.h:65-360 (exact FK),
// :553-731 (analytic Jacobian), :783-1234 (mutation approximator), reading the flat model of
// include/bioik_hip.h instead of moveit::core::RobotModel (MoveIt is not on disk).
#pragma once
#include <atomic>
#include <algorithm>
#include <cfloat>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/bioik_hip.h"
#include "../bio_ik_amd/csrc/bioik_sincos.h"
#include "orc_math.h"

namespace orc {

// Reference quirks.  0 (default): the two deliberate fixes the device shares — Q1: computeApproximateMutation1 writes the
// unchanged input frame for tips a variable does not move; Q4: children are pre-selected in a STABLE order.
// 1: literal reference behaviour — Q1: such tips keep whatever the output buffer held (forward_kinematics.h:1017 `continue`),
// Q4: std::sort (ik_evolution_2.cpp:376).  Mode 1 exists to pin this restatement, trajectory by trajectory, against the
// reference's own code compiled into oracle/_ref (tests/test_oracle_vs_reference.py).
inline int& quirk_mode() {
    static int mode = 0;
    return mode;
}
// diagnostics (tests, tools/robot_fuzz_hostsim.py): how many candidates of the memetic line search had a gene of magnitude >= 1e300 -- an infinite step (curvature
// 0, slope not) on a joint WITHOUT limits, clipped to +-DBL_MAX.  Quirk Q7 (orc_evolution.h): no candidate in the default mode, literal in mode 1.
inline std::atomic<unsigned long long>& unbounded_candidates() {
    static std::atomic<unsigned long long> n{0};
    return n;
}

struct Link {
    int parent;
    Frame origin;  // link_frames[link] = LinkModel::getJointOriginTransform(), forward_kinematics.h:201-203
    int type;
    Vec3 axis;     // joint_axis_list, forward_kinematics.h:205-212
    int first_var;
    int var_count;
    int mimic;     // link index of the mimicked joint or -1
    double mimic_factor, mimic_offset;
    double mass;   // urdf::Link::inertial->mass (0: no inertial), goal_types.cpp:236-247
    Vec3 center;   // urdf::Link::inertial->origin.position
};

// robot_info.h:48-55
struct VarInfo {
    double clip_min, clip_max, span, min, max, max_velocity, max_velocity_rcp;
};

inline int joint_var_count(int type) {
    switch (type) {
        case BIOIK_JOINT_FIXED: return 0;
        case BIOIK_JOINT_REVOLUTE: return 1;
        case BIOIK_JOINT_PRISMATIC: return 1;
        case BIOIK_JOINT_FLOATING: return 7;
        case BIOIK_JOINT_PLANAR: return 3;
    }
    throw std::runtime_error("unknown joint type");
}

struct Model {
    std::vector<Link> links;
    std::vector<VarInfo> vars;
    std::vector<int> var_joint;     // RobotModel::getJointOfVariable
    std::vector<int> mimic_joints;  // RobotModel::getMimicJointModels(), model order

    explicit Model(const bioik_model_desc& d) {
        if (d.n_links == 0) throw std::runtime_error("model has no links");
        links.resize(d.n_links);
        var_joint.assign(d.n_variables, -1);
        for (uint32_t i = 0; i < d.n_links; i++) {
            Link& l = links[i];
            l.parent = d.link_parent[i];
            if (l.parent >= (int)i) throw std::runtime_error("links must be ordered parent before child");
            l.origin = frame_from7(d.link_origin + 7 * i);
            l.type = d.joint_type[i];
            l.axis = {d.joint_axis[3 * i], d.joint_axis[3 * i + 1], d.joint_axis[3 * i + 2]};
            l.first_var = d.joint_first_variable[i];
            l.var_count = joint_var_count(l.type);
            l.mimic = d.joint_mimic ? d.joint_mimic[i] : -1;
            l.mimic_factor = d.joint_mimic_factor ? d.joint_mimic_factor[i] : 1.0;
            l.mimic_offset = d.joint_mimic_offset ? d.joint_mimic_offset[i] : 0.0;
            l.mass = d.link_mass ? d.link_mass[i] : 0.0;
            l.center = d.link_center ? Vec3{d.link_center[3 * i], d.link_center[3 * i + 1], d.link_center[3 * i + 2]} : Vec3{0.0, 0.0, 0.0};
            if (l.var_count > 0) {
                if (l.first_var < 0 || l.first_var + l.var_count > (int)d.n_variables) throw std::runtime_error("joint variable index out of range");
                for (int v = 0; v < l.var_count; v++) var_joint[l.first_var + v] = (int)i;
            }
            if (l.mimic >= 0) mimic_joints.push_back((int)i);
        }
        // RobotInfo(model): robot_info.h:70-106
        vars.resize(d.n_variables);
        for (uint32_t v = 0; v < d.n_variables; v++) {
            VarInfo info;
            bool bounded = d.var_bounded[v] != 0;
            int j = var_joint[v];
            if (j >= 0 && links[j].type == BIOIK_JOINT_REVOLUTE)
                if (d.var_max[v] - d.var_min[v] >= 2 * M_PI * 0.9999) bounded = false;  // :82-84
            info.min = d.var_min[v];
            info.max = d.var_max[v];
            info.clip_min = bounded ? info.min : -DBL_MAX;
            info.clip_max = bounded ? info.max : +DBL_MAX;
            info.span = info.max - info.min;
            if (!(info.span >= 0 && info.span < FLT_MAX)) info.span = 1;  // :94
            info.max_velocity = d.var_max_velocity[v];
            info.max_velocity_rcp = info.max_velocity > 0.0 ? 1.0 / info.max_velocity : 0.0;
            vars[v] = info;
        }
    }
    // robot_info.h:109-113 (clamp2)
    inline double clip(double p, size_t i) const {
        const VarInfo& info = vars[i];
        if (p < info.clip_min) p = info.clip_min;
        if (p > info.clip_max) p = info.clip_max;
        return p;
    }
    inline bool is_revolute(size_t v) const { return var_joint[v] >= 0 && links[var_joint[v]].type == BIOIK_JOINT_REVOLUTE; }
};

// RobotJointEvaluator + RobotFK_Fast_Base + RobotFK_Jacobian + RobotFK_Mutator
struct RobotFK {
    const Model* model;
    std::vector<int> tip_links;
    std::vector<int> link_schedule;
    std::vector<Frame> tip_frames, global_frames;
    std::vector<double> variables;
    std::vector<std::vector<int>> joint_dependencies;
    std::vector<int> tip_dependencies;
    // mutator state (forward_kinematics.h:786-794)
    std::vector<double> approx_jacobian;  // [6T][D] row-major (Eigen::MatrixXd in the reference)
    size_t approx_cols = 0;
    std::vector<std::vector<Frame>> approx_frames;  // [tip][robot variable]
    std::vector<size_t> approx_variable_indices;
    std::vector<std::vector<int>> approx_mask;     // [tip][robot variable]
    std::vector<std::vector<size_t>> approx_map;   // [tip] -> gene indices that move the tip

    explicit RobotFK(const Model* m) : model(m) {}

    // forward_kinematics.h:78-139
    void get_joint_frame(int joint, const double* vars, Frame& frame) const {
        const Link& l = model->links[joint];
        switch (l.type) {
            case BIOIK_JOINT_FIXED: frame = identity_frame(); return;
            case BIOIK_JOINT_REVOLUTE: {
                double v = vars[l.first_var];
                double half_angle = v * 0.5;
                double fcos, fsin;
                if (fused()) {
                    bioik_sincos(half_angle, &fsin, &fcos);
                } else {
                    fcos = std::cos(half_angle);
                    fsin = std::sin(half_angle);
                }
                frame = Frame{{0.0, 0.0, 0.0}, {l.axis.x * fsin, l.axis.y * fsin, l.axis.z * fsin, fcos}};
                return;
            }
            case BIOIK_JOINT_PRISMATIC: {
                double v = vars[l.first_var];
                frame = Frame{l.axis * v, {0.0, 0.0, 0.0, 1.0}};
                return;
            }
            case BIOIK_JOINT_FLOATING: {
                const double* vv = vars + l.first_var;
                frame.pos = {vv[0], vv[1], vv[2]};
                frame.rot = normalized(Quat{vv[3], vv[4], vv[5], vv[6]});
                return;
            }
            default: {
                // :128-135 JointModel::computeTransform; MoveIt PlanarJointModel: Translation(x,y,0)*AngleAxis(theta,Z)
                const double* vv = vars + l.first_var;
                double h = vv[2] * 0.5;
                double hs, hc;
                if (fused()) {
                    bioik_sincos(h, &hs, &hc);
                } else {
                    hs = std::sin(h), hc = std::cos(h);
                }
                frame.pos = {vv[0], vv[1], 0.0};
                frame.rot = {0.0, 0.0, hs, hc};
                return;
            }
        }
    }

    // forward_kinematics.h:230-246
    void update_mimic(std::vector<double>& values) const {
        for (int j : model->mimic_joints) {
            const Link& l = model->links[j];
            if (l.var_count == 0) continue;
            int src = model->links[l.mimic].first_var;
            int dest = l.first_var;
            values[dest] = values[src] * l.mimic_factor + l.mimic_offset;
        }
    }

    // forward_kinematics.h:253-330 + 566-599
    void initialize(const std::vector<int>& tip_link_indices) {
        tip_links = tip_link_indices;
        size_t tip_count = tip_links.size();
        tip_frames.assign(tip_count, identity_frame());
        global_frames.assign(model->links.size(), identity_frame());
        link_schedule.clear();
        for (int tip : tip_links) {
            std::vector<int> chain;
            for (int link = tip; link >= 0; link = model->links[link].parent) chain.push_back(link);
            std::reverse(chain.begin(), chain.end());
            for (int link : chain) {
                if (std::find(link_schedule.begin(), link_schedule.end(), link) != link_schedule.end()) continue;
                link_schedule.push_back(link);
            }
        }
        joint_dependencies.assign(model->links.size(), {});
        for (int link : link_schedule) joint_dependencies[link].push_back(link);
        for (int link : link_schedule) {
            int mimic = model->links[link].mimic;
            if (mimic >= 0) {
                while (model->links[mimic].mimic >= 0 && model->links[mimic].mimic != link) mimic = model->links[mimic].mimic;
                joint_dependencies[mimic].push_back(link);
            }
        }
        tip_dependencies.assign(model->links.size() * tip_count, 0);
        for (size_t t = 0; t < tip_count; t++)
            for (int link = tip_links[t]; link >= 0; link = model->links[link].parent) tip_dependencies[link * tip_count + t] = 1;
    }

    // forward_kinematics.h:331-354
    void apply_configuration(const std::vector<double>& jj0) {
        variables = jj0;
        update_mimic(variables);
        Frame jf;
        for (int link : link_schedule) {
            const Link& l = model->links[link];
            get_joint_frame(link, variables.data(), jf);
            if (l.parent >= 0)
                concat(global_frames[l.parent], l.origin, jf, global_frames[link]);
            else
                concat(l.origin, jf, global_frames[link]);
        }
        for (size_t t = 0; t < tip_links.size(); t++) tip_frames[t] = global_frames[tip_links[t]];
    }

    // forward_kinematics.h:600-730; jacobian(row, col) stored row-major [6T][cols]
    void compute_jacobian(const std::vector<size_t>& variable_indices) {
        const double step_size = 0.00001;
        const double inv_step_size = 1.0 / step_size;
        size_t tip_count = tip_frames.size();
        size_t cols = variable_indices.size();
        approx_cols = cols;
        approx_jacobian.assign(tip_count * 6 * cols, 0.0);
        auto J = [&](size_t row, size_t col) -> double& { return approx_jacobian[row * cols + col]; };
        for (size_t icol = 0; icol < cols; icol++) {
            size_t ivar = variable_indices[icol];
            int var_joint = model->var_joint[ivar];
            if (model->links[var_joint].mimic >= 0) continue;  // :623
            for (int joint : joint_dependencies[var_joint]) {
                double scale = 1;
                for (int m = joint; model->links[m].mimic >= 0 && model->links[m].mimic != joint; m = model->links[m].mimic)
                    scale *= model->links[m].mimic_factor;  // :626-630
                const Link& jl = model->links[joint];
                int link = joint;  // child link of the joint
                switch (jl.type) {
                    case BIOIK_JOINT_FIXED: continue;
                    case BIOIK_JOINT_REVOLUTE: {
                        const Frame& link_frame = global_frames[link];
                        for (size_t itip = 0; itip < tip_count; itip++) {
                            if (!tip_dependencies[joint * tip_count + itip]) continue;
                            const Frame& tip_frame = tip_frames[itip];
                            Quat q = tf2_mul(inverse(link_frame.rot), tip_frame.rot);  // :648
                            q = inverse(q);
                            Vec3 rot = jl.axis;
                            quat_mul_vec(q, rot, rot);
                            Vec3 vel = link_frame.pos - tip_frame.pos;
                            quat_mul_vec(inverse(tip_frame.rot), vel, vel);
                            vel = cross(vel, rot);
                            J(itip * 6 + 0, icol) += vel.x * scale;
                            J(itip * 6 + 1, icol) += vel.y * scale;
                            J(itip * 6 + 2, icol) += vel.z * scale;
                            J(itip * 6 + 3, icol) += rot.x * scale;
                            J(itip * 6 + 4, icol) += rot.y * scale;
                            J(itip * 6 + 5, icol) += rot.z * scale;
                        }
                        continue;
                    }
                    case BIOIK_JOINT_PRISMATIC: {
                        const Frame& link_frame = global_frames[link];
                        for (size_t itip = 0; itip < tip_count; itip++) {
                            if (!tip_dependencies[joint * tip_count + itip]) continue;
                            const Frame& tip_frame = tip_frames[itip];
                            Quat q = tf2_mul(inverse(link_frame.rot), tip_frame.rot);
                            q = inverse(q);
                            Vec3 v;
                            quat_mul_vec(q, jl.axis, v);
                            J(itip * 6 + 0, icol) += v.x * scale;
                            J(itip * 6 + 1, icol) += v.y * scale;
                            J(itip * 6 + 2, icol) += v.z * scale;
                        }
                        continue;
                    }
                    default: {
                        // :695-726 numeric differentiation for the remaining joint types
                        size_t ivar2 = ivar;
                        if (jl.mimic >= 0) ivar2 = ivar2 - model->links[var_joint].first_var + jl.first_var;
                        Frame link_frame_1 = global_frames[link];
                        double v0 = variables[ivar2];
                        variables[ivar2] = v0 + step_size;
                        Frame joint_frame_2;
                        get_joint_frame(joint, variables.data(), joint_frame_2);
                        variables[ivar2] = v0;
                        Frame link_frame_2;
                        if (jl.parent >= 0)
                            concat(global_frames[jl.parent], jl.origin, joint_frame_2, link_frame_2);
                        else
                            concat(jl.origin, joint_frame_2, link_frame_2);
                        for (size_t itip = 0; itip < tip_count; itip++) {
                            if (!tip_dependencies[joint * tip_count + itip]) continue;
                            Frame tip_frame_1 = tip_frames[itip];
                            Frame tip_frame_2;
                            change(link_frame_2, link_frame_1, tip_frame_1, tip_frame_2);
                            double tw[6];
                            frame_twist(tip_frame_1, tip_frame_2, tw);
                            for (int k = 0; k < 6; k++) J(itip * 6 + k, icol) += tw[k] * inv_step_size * scale;
                        }
                        continue;
                    }
                }
            }
        }
    }

    // forward_kinematics.h:802-930
    void initialize_mutation_approximator(const std::vector<size_t>& variable_indices) {
        approx_variable_indices = variable_indices;
        size_t tip_count = tip_links.size();
        size_t nvar = model->vars.size();
        if (approx_frames.size() < tip_count) approx_frames.resize(tip_count);
        for (size_t t = 0; t < tip_count; t++) approx_frames[t].resize(nvar, identity_frame());
        for (size_t t = 0; t < tip_count; t++)
            for (size_t ivar : variable_indices) approx_frames[t][ivar] = identity_frame();
        compute_jacobian(variable_indices);
        size_t cols = variable_indices.size();
        for (size_t icol = 0; icol < cols; icol++) {
            size_t ivar = variable_indices[icol];
            for (size_t t = 0; t < tip_count; t++) {
                {
                    Vec3 tv = {approx_jacobian[(t * 6 + 0) * cols + icol], approx_jacobian[(t * 6 + 1) * cols + icol],
                               approx_jacobian[(t * 6 + 2) * cols + icol]};
                    quat_mul_vec(tip_frames[t].rot, tv, tv);
                    approx_frames[t][ivar].pos = tv;
                }
                {
                    Quat q = {approx_jacobian[(t * 6 + 3) * cols + icol] * 0.5, approx_jacobian[(t * 6 + 4) * cols + icol] * 0.5,
                              approx_jacobian[(t * 6 + 5) * cols + icol] * 0.5, 1.0};
                    quat_mul_quat(tip_frames[t].rot, q, q);
                    q = q - tip_frames[t].rot;
                    approx_frames[t][ivar].rot = q;
                }
            }
        }
        if (approx_mask.size() < tip_count) approx_mask.resize(tip_count);
        if (approx_map.size() < tip_count) approx_map.resize(tip_count);
        for (size_t t = 0; t < tip_count; t++) {
            if (approx_mask[t].size() < nvar) approx_mask[t].resize(nvar);
            approx_map[t].clear();
            for (size_t ii = 0; ii < variable_indices.size(); ii++) {
                size_t ivar = variable_indices[ii];
                const Frame& f = approx_frames[t][ivar];
                bool b = false;
                b |= (f.pos.x != 0.0);
                b |= (f.pos.y != 0.0);
                b |= (f.pos.z != 0.0);
                b |= (f.rot.x != 0.0);
                b |= (f.rot.y != 0.0);
                b |= (f.rot.z != 0.0);
                approx_mask[t][ivar] = b;
                if (b) approx_map[t].push_back(ii);
            }
        }
    }

    // forward_kinematics.h:1009-1058 (scalar variant).  Deliberate fix: the
    // reference leaves output[itip] untouched (stale) for tips the variable does not move; the evidently
    // intended value — the unchanged input frame — is written here.
    void compute_approximate_mutation1(size_t variable_index, double variable_delta, const std::vector<Frame>& input,
                                       std::vector<Frame>& output) const {
        size_t tip_count = tip_links.size();
        output.resize(tip_count);
        for (size_t t = 0; t < tip_count; t++) {
            if (approx_mask[t][variable_index] == 0) {
                if (quirk_mode() == 0) output[t] = input[t];  // mode 1: stale, as in the reference
                continue;
            }
            const Frame& jd = approx_frames[t][variable_index];
            const Frame& tf = input[t];
            double px = tf.pos.x, py = tf.pos.y, pz = tf.pos.z;
            double rx = tf.rot.x, ry = tf.rot.y, rz = tf.rot.z, rw = tf.rot.w;
            px = madd(jd.pos.x, variable_delta, px);
            py = madd(jd.pos.y, variable_delta, py);
            pz = madd(jd.pos.z, variable_delta, pz);
            rx = madd(jd.rot.x, variable_delta, rx);
            ry = madd(jd.rot.y, variable_delta, ry);
            rz = madd(jd.rot.z, variable_delta, rz);
            rw = madd(jd.rot.w, variable_delta, rw);
            output[t] = Frame{{px, py, pz}, {rx, ry, rz, rw}};
        }
    }

    // forward_kinematics.h:1175-1233 (scalar variant); out[m*T + t]
    void compute_approximate_mutations(size_t mutation_count, const double* const* mutation_values, Frame* out) const {
        const double* p_variables = variables.data();
        size_t tip_count = tip_links.size();
        for (size_t t = 0; t < tip_count; t++) {
            const std::vector<Frame>& joint_deltas = approx_frames[t];
            const Frame& tf = tip_frames[t];
            for (size_t m = 0; m < mutation_count; m++) {
                double px = tf.pos.x, py = tf.pos.y, pz = tf.pos.z;
                double rx = tf.rot.x, ry = tf.rot.y, rz = tf.rot.z, rw = tf.rot.w;
                for (size_t vii : approx_map[t]) {
                    size_t variable_index = approx_variable_indices[vii];
                    double variable_delta = mutation_values[m][vii] - p_variables[variable_index];
                    px = madd(joint_deltas[variable_index].pos.x, variable_delta, px);
                    py = madd(joint_deltas[variable_index].pos.y, variable_delta, py);
                    pz = madd(joint_deltas[variable_index].pos.z, variable_delta, pz);
                    rx = madd(joint_deltas[variable_index].rot.x, variable_delta, rx);
                    ry = madd(joint_deltas[variable_index].rot.y, variable_delta, ry);
                    rz = madd(joint_deltas[variable_index].rot.z, variable_delta, rz);
                    rw = madd(joint_deltas[variable_index].rot.w, variable_delta, rw);
                }
                out[m * tip_count + t] = Frame{{px, py, pz}, {rx, ry, rz, rw}};
            }
        }
    }
};

}  // namespace orc
