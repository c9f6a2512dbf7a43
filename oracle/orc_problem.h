// orc_problem.h — ORACLE (test infrastructure): Problem + built-in goal costs.
//
// Restates reference src/problem.cpp:57-342 (initialize, computeGoalFitness, checkSolutionActiveVariables),
// include/bio_ik/goal.h:49-119 (GoalContext accessors) and every closed-form Goal::evaluate of
// include/bio_ik/goal_types.h:80-712.  The goal structure comes from the problem template of
// include/bioik_hip.h; the per-query numbers (initial_guess, goal parameters) from a Query.
#pragma once
#include <climits>
#include <cmath>
#include <stdexcept>
#include <vector>

#include "orc_model.h"

namespace orc {

inline int goal_param_count(int type) {
    switch (type) {
        case BIOIK_GOAL_POSITION: return 3;
        case BIOIK_GOAL_ORIENTATION: return 4;
        case BIOIK_GOAL_POSE: return 8;
        case BIOIK_GOAL_LOOK_AT: return 6;
        case BIOIK_GOAL_MAX_DISTANCE: return 4;
        case BIOIK_GOAL_MIN_DISTANCE: return 4;
        case BIOIK_GOAL_LINE: return 6;
        case BIOIK_GOAL_PLANE: return 6;
        case BIOIK_GOAL_AVOID_JOINT_LIMITS: return 0;
        case BIOIK_GOAL_CENTER_JOINTS: return 0;
        case BIOIK_GOAL_REGULARIZATION: return 0;
        case BIOIK_GOAL_MINIMAL_DISPLACEMENT: return 0;
        case BIOIK_GOAL_JOINT_VARIABLE: return 1;
        case BIOIK_GOAL_SIDE: return 6;
        case BIOIK_GOAL_DIRECTION: return 6;
        case BIOIK_GOAL_CONE: return 11;
        case BIOIK_GOAL_BALANCE: return 6;
    }
    return -1;
}

struct Query {
    const double* initial_guess;  // Problem::initial_guess [V]
    const double* params;         // [P]
};

// Problem::GoalInfo (problem.h:121-130) without the per-query numbers
struct GoalInfo {
    int type;
    long tip_index;        // goal_link_indices_[0] (problem tip index), -1 when the goal has no link
    long variable_index;   // goal_variable_indices_[0]: >=0 gene index, <0 = -1-ivar (fixed joint), LONG_MIN none
    double weight, weight_sq;
    bool secondary;
    int param_offset;
    // BalanceGoal (goal_types.h:544-549): one entry per link with mass, in model link order -- its problem tip index, its centre of mass
    // in the link frame and its share of the total mass
    std::vector<long> balance_tips;
    std::vector<Vec3> balance_centers;
    std::vector<double> balance_weights;
};

struct Problem {
    const Model* model;
    std::vector<int> group_joints;
    std::vector<char> group_variable;  // membership of a variable in joint_model_group->getVariableNames()
    std::vector<char> fixed_joint;
    std::vector<size_t> active_variables;
    std::vector<int> tip_link_indices;
    std::vector<long> link_tip_indices;
    std::vector<GoalInfo> goals, secondary_goals;
    std::vector<double> minimal_displacement_factors;
    int param_count = 0;

    // problem.cpp:57-65
    size_t add_tip_link(int link) {
        if (link_tip_indices[link] < 0) {
            link_tip_indices[link] = (long)tip_link_indices.size();
            tip_link_indices.push_back(link);
        }
        return (size_t)link_tip_indices[link];
    }
    // problem.cpp:103-126
    long add_active_variable(int ivar) {
        int joint = model->var_joint[ivar];
        if (fixed_joint[joint]) return (long)-1 - (long)ivar;
        for (size_t i = 0; i < active_variables.size(); i++)
            if (active_variables[i] == (size_t)ivar) return (long)i;
        if (group_variable[ivar]) {
            active_variables.push_back((size_t)ivar);
            return (long)active_variables.size() - 1;
        }
        throw std::runtime_error("joint variable not found");
    }

    // problem.cpp:72-228
    Problem(const Model* m, const bioik_problem_desc& d) : model(m) {
        size_t nl = m->links.size(), nv = m->vars.size();
        group_joints.assign(d.group_joints, d.group_joints + d.n_group_joints);
        group_variable.assign(nv, 0);
        for (int j : group_joints) {
            if (j < 0 || j >= (int)nl) throw std::runtime_error("group joint out of range");
            const Link& l = m->links[j];
            for (int v = 0; v < l.var_count; v++) group_variable[l.first_var + v] = 1;
        }
        fixed_joint.assign(nl, 0);
        for (uint32_t i = 0; i < d.n_fixed_joints; i++) {
            int j = d.fixed_joints[i];
            if (j < 0 || j >= (int)nl) throw std::runtime_error("fixed joint out of range");
            fixed_joint[j] = 1;
        }
        link_tip_indices.assign(nl, -1);
        for (uint32_t gi = 0; gi < d.n_goals; gi++) {
            const bioik_goal_desc& g = d.goals[gi];
            GoalInfo info;
            info.type = g.type;
            int np = goal_param_count(g.type);
            if (np < 0) throw std::runtime_error("unknown goal type");
            info.tip_index = -1;
            info.variable_index = LONG_MIN;
            if (g.link >= 0) {
                if (g.link >= (int)nl) throw std::runtime_error("link not found");
                info.tip_index = (long)add_tip_link(g.link);
            }
            if (g.variable >= 0) {
                if (g.variable >= (int)nv) throw std::runtime_error("joint variable not found");
                info.variable_index = add_active_variable(g.variable);
            }
            if (g.type == BIOIK_GOAL_BALANCE) {  // BalanceGoal::describe, goal_types.cpp:231-255
                double total = 0.0;
                for (size_t l = 0; l < nl; l++) {
                    const double mass = m->links[l].mass;
                    if (!(mass > 0)) continue;
                    info.balance_centers.push_back(m->links[l].center);
                    info.balance_weights.push_back(mass);
                    total += mass;
                    info.balance_tips.push_back((long)add_tip_link((int)l));
                }
                for (double& w : info.balance_weights) w /= total;
            }
            info.weight = g.weight;
            info.weight_sq = info.weight * info.weight;
            info.secondary = g.secondary != 0;
            info.param_offset = param_count;
            param_count += np;
            if (info.secondary)
                secondary_goals.push_back(info);
            else
                goals.push_back(info);
        }
        // active variables from the active subtree, problem.cpp:191-204
        std::vector<int> joint_usage(nl, 0);
        for (int tip : tip_link_indices)
            for (int link = tip; link >= 0; link = m->links[link].parent) joint_usage[link] = 1;
        for (size_t j = 0; j < nl; j++)
            if (fixed_joint[j]) joint_usage[j] = 0;
        for (int j : group_joints) {
            const Link& l = m->links[j];
            if (joint_usage[j] && l.mimic < 0)
                for (int v = 0; v < l.var_count; v++) add_active_variable(l.first_var + v);
        }
        // problem.cpp:206-225
        minimal_displacement_factors.resize(active_variables.size());
        double s = 0;
        for (size_t ivar : active_variables) s += m->vars[ivar].max_velocity_rcp;
        if (s > 0) {
            for (size_t i = 0; i < active_variables.size(); i++)
                minimal_displacement_factors[i] = m->vars[active_variables[i]].max_velocity_rcp / s;
        } else {
            for (size_t i = 0; i < active_variables.size(); i++) minimal_displacement_factors[i] = 1.0 / active_variables.size();
        }
    }

    // GoalContext::getVariablePosition, goal.h:70-77
    inline double goal_variable_position(const GoalInfo& g, const Query& q, const double* genes) const {
        long j = g.variable_index;
        if (j >= 0) return genes[j];
        return q.initial_guess[-1 - j];
    }

    // Goal::evaluate of goal_types.h, by opcode
    double evaluate(const GoalInfo& g, const Query& q, const Frame* tip_frames, const double* genes) const {
        const double* P = q.params + g.param_offset;
        const Frame* fbp = g.tip_index >= 0 ? &tip_frames[g.tip_index] : nullptr;
        switch (g.type) {
            case BIOIK_GOAL_POSITION:  // goal_types.h:96
                return distance2(fbp->pos, Vec3{P[0], P[1], P[2]});
            case BIOIK_GOAL_ORIENTATION: {  // :115-124
                Quat o = {P[0], P[1], P[2], P[3]};
                return std::fmin(length2(o - fbp->rot), length2(o + fbp->rot));
            }
            case BIOIK_GOAL_POSE: {  // :149-180
                double e = 0.0;
                e += distance2(fbp->pos, Vec3{P[0], P[1], P[2]});
                Quat o = {P[3], P[4], P[5], P[6]};
                double rs = P[7];
                e += std::fmin(length2(o - fbp->rot), length2(o + fbp->rot)) * (rs * rs);
                return e;
            }
            case BIOIK_GOAL_LOOK_AT: {  // :204-211
                Vec3 axis;
                quat_mul_vec(fbp->rot, Vec3{P[0], P[1], P[2]}, axis);
                Vec3 target = {P[3], P[4], P[5]};
                return distance2(normalized(target - fbp->pos), normalized(axis));
            }
            case BIOIK_GOAL_MAX_DISTANCE: {  // :235-240
                double d = std::fmax(0.0, distance(fbp->pos, Vec3{P[0], P[1], P[2]}) - P[3]);
                return d * d;
            }
            case BIOIK_GOAL_MIN_DISTANCE: {  // :264-269
                double d = std::fmax(0.0, P[3] - distance(fbp->pos, Vec3{P[0], P[1], P[2]}));
                return d * d;
            }
            case BIOIK_GOAL_LINE: {  // :293-297
                Vec3 position = {P[0], P[1], P[2]}, direction = {P[3], P[4], P[5]};
                return distance2(position, fbp->pos - direction * dot(direction, fbp->pos - position));
            }
            case BIOIK_GOAL_PLANE: {  // :321-327
                Vec3 position = {P[0], P[1], P[2]}, normal = {P[3], P[4], P[5]};
                double signed_dist = dot(fbp->pos - position, normal);
                return signed_dist * signed_dist;
            }
            case BIOIK_GOAL_AVOID_JOINT_LIMITS: {  // :387-401
                double sum = 0.0;
                for (size_t i = 0; i < active_variables.size(); i++) {
                    const VarInfo& info = model->vars[active_variables[i]];
                    if (info.clip_max == DBL_MAX) continue;
                    double d = genes[i] - (info.min + info.max) * 0.5;
                    d = std::fmax(0.0, std::fabs(d) * 2.0 - info.span * 0.5);
                    d *= minimal_displacement_factors[i];
                    sum += d * d;
                }
                return sum;
            }
            case BIOIK_GOAL_CENTER_JOINTS: {  // :412-425
                double sum = 0.0;
                for (size_t i = 0; i < active_variables.size(); i++) {
                    const VarInfo& info = model->vars[active_variables[i]];
                    if (info.clip_max == DBL_MAX) continue;
                    double d = genes[i] - (info.min + info.max) * 0.5;
                    d *= minimal_displacement_factors[i];
                    sum += d * d;
                }
                return sum;
            }
            case BIOIK_GOAL_REGULARIZATION: {  // :435-444
                double sum = 0.0;
                for (size_t i = 0; i < active_variables.size(); i++) {
                    double d = genes[i] - q.initial_guess[active_variables[i]];
                    sum += d * d;
                }
                return sum;
            }
            case BIOIK_GOAL_MINIMAL_DISPLACEMENT: {  // :455-465
                double sum = 0.0;
                for (size_t i = 0; i < active_variables.size(); i++) {
                    double d = genes[i] - q.initial_guess[active_variables[i]];
                    d *= minimal_displacement_factors[i];
                    sum += d * d;
                }
                return sum;
            }
            case BIOIK_GOAL_JOINT_VARIABLE: {  // :494-498
                double d = P[0] - goal_variable_position(g, q, genes);
                return d * d;
            }
            case BIOIK_GOAL_SIDE: {  // :606-613
                Vec3 v;
                quat_mul_vec(fbp->rot, Vec3{P[0], P[1], P[2]}, v);
                double f = std::fmax(0.0, dot(v, Vec3{P[3], P[4], P[5]}));
                return f * f;
            }
            case BIOIK_GOAL_DIRECTION: {  // :637-643
                Vec3 v;
                quat_mul_vec(fbp->rot, Vec3{P[0], P[1], P[2]}, v);
                return distance2(v, Vec3{P[3], P[4], P[5]});
            }
            case BIOIK_GOAL_BALANCE: {  // goal_types.cpp:257-272
                Vec3 center = {0.0, 0.0, 0.0};
                for (size_t i = 0; i < g.balance_tips.size(); i++) {
                    const Frame& frame = tip_frames[g.balance_tips[i]];
                    Vec3 c = g.balance_centers[i];
                    quat_mul_vec(frame.rot, c, c);
                    c = c + frame.pos;
                    center = center + c * g.balance_weights[i];
                }
                Vec3 target = {P[0], P[1], P[2]}, axis = {P[3], P[4], P[5]};
                center = center - target;
                center = center - axis * dot(axis, center);
                return length2(center);
            }
            case BIOIK_GOAL_CONE: {  // :700-711
                double sum = 0.0;
                Vec3 v;
                quat_mul_vec(fbp->rot, Vec3{P[4], P[5], P[6]}, v);
                double d = std::fmax(0.0, angle(v, Vec3{P[7], P[8], P[9]}) - P[10]);
                sum += d * d;
                double w = P[3];
                sum += w * w * length2(Vec3{P[0], P[1], P[2]} - fbp->pos);
                return sum;
            }
        }
        throw std::runtime_error("unknown goal type");
    }

    // problem.cpp:244-257
    double compute_goal_fitness(const std::vector<GoalInfo>& gl, const Query& q, const Frame* tip_frames, const double* genes) const {
        double sum = 0.0;
        for (const GoalInfo& g : gl) sum += evaluate(g, q, tip_frames, genes) * g.weight_sq;
        return sum;
    }

    // problem.cpp:259-341 ; dpos/drot/dtwist already normalised as in problem.cpp:90-95
    bool check_solution(const Query& q, const Frame* tip_frames, const double* genes, double dpos, double drot, double dtwist) const {
        for (const GoalInfo& goal : goals) {
            switch (goal.type) {
                case BIOIK_GOAL_POSITION: {
                    const double* P = q.params + goal.param_offset;
                    Frame fa = identity_frame();
                    fa.pos = {P[0], P[1], P[2]};
                    const Frame& fb = tip_frames[goal.tip_index];
                    if (dpos != DBL_MAX) {
                        double p_dist = length(fb.pos - fa.pos);
                        if (!(p_dist <= dpos)) return false;
                    }
                    if (dtwist != DBL_MAX) {
                        double tw[6];
                        kdl_pose_twist(fa, fb, tw);
                        for (int k = 0; k < 3; k++)
                            if (!(std::fabs(tw[k]) < dtwist)) return false;  // KDL::Equal(Vector, Vector, eps)
                    }
                    continue;
                }
                case BIOIK_GOAL_ORIENTATION: {
                    const double* P = q.params + goal.param_offset;
                    Frame fa = identity_frame();
                    fa.rot = {P[0], P[1], P[2], P[3]};
                    const Frame& fb = tip_frames[goal.tip_index];
                    if (drot != DBL_MAX) {
                        double r_dist = angle_shortest_path(fb.rot, fa.rot);
                        r_dist = r_dist * 180 / M_PI;
                        if (!(r_dist <= drot)) return false;
                    }
                    if (dtwist != DBL_MAX) {
                        double tw[6];
                        kdl_pose_twist(fa, fb, tw);
                        for (int k = 3; k < 6; k++)
                            if (!(std::fabs(tw[k]) < dtwist)) return false;
                    }
                    continue;
                }
                case BIOIK_GOAL_POSE: {
                    const double* P = q.params + goal.param_offset;
                    Frame fa = {{P[0], P[1], P[2]}, {P[3], P[4], P[5], P[6]}};
                    const Frame& fb = tip_frames[goal.tip_index];
                    if (dpos != DBL_MAX || drot != DBL_MAX) {
                        double p_dist = length(fb.pos - fa.pos);
                        double r_dist = angle_shortest_path(fb.rot, fa.rot);
                        r_dist = r_dist * 180 / M_PI;
                        if (!(p_dist <= dpos)) return false;
                        if (!(r_dist <= drot)) return false;
                    }
                    if (dtwist != DBL_MAX) {
                        double tw[6];
                        kdl_pose_twist(fa, fb, tw);
                        for (int k = 0; k < 6; k++)
                            if (!(std::fabs(tw[k]) < dtwist)) return false;
                    }
                    continue;
                }
                default: {
                    double dmax = DBL_MAX;
                    dmax = std::fmin(dmax, dpos);
                    dmax = std::fmin(dmax, dtwist);
                    double d = evaluate(goal, q, tip_frames, genes) * goal.weight_sq;
                    if (!(d < dmax * dmax)) return false;
                }
            }
        }
        return true;
    }
};

// problem.cpp:90-95
inline double normalize_threshold(double v) {
    if (v < 0.0 || v >= FLT_MAX || !std::isfinite(v)) return DBL_MAX;
    return v;
}

}  // namespace orc
