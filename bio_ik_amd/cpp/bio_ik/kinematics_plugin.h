// bio_ik/kinematics_plugin.h — `bio_ik_kinematics_plugin::BioIKKinematicsPlugin` for programs WITHOUT MoveIt.
//
// The reference plugin's interface (src/kinematics_plugin.cpp:117-671: same method names, argument meaning and error behaviour as its
// kinematics::KinematicsBase implementation) over `bio_ik::RobotModel` (bio_ik/robot_model.h, bio_ik/urdf.h) instead of
// moveit::core::RobotModel, with light stand-ins for the two ROS message types.  Everything the plugin does around the solver call is
// the shared implementation of bio_ik/plugin_core.h; this class only fills a ModelView from its model type and converts poses.  The
// MoveIt build of the same plugin is src/kinematics_plugin_hip.cpp.  Header-only; link with libbioik_hip.so.
#pragma once
#include "bio_ik.h"
#include "plugin_core.h"
#include "robot_model.h"

namespace geometry_msgs {
struct Pose {
    struct { double x = 0, y = 0, z = 0; } position;
    struct { double x = 0, y = 0, z = 0, w = 1; } orientation;
};
}  // namespace geometry_msgs
namespace moveit_msgs {
struct MoveItErrorCodes {
    enum { SUCCESS = 1, NO_IK_SOLUTION = -31 };
    int val = 0;
};
}  // namespace moveit_msgs

namespace bio_ik_kinematics_plugin {

typedef std::function<void(const geometry_msgs::Pose&, const std::vector<double>&, moveit_msgs::MoveItErrorCodes&)> IKCallbackFn;

struct BioIKParams : bio_ik::core::Settings {  // the kinematics.yaml keys of the reference (kinematics_plugin.cpp:243-328) + the additive gpu_* keys
    double rotation_scale = 0.5;
    bool position_only_ik = false;
    double center_joints_weight = 0, avoid_joint_limits_weight = 0, minimal_displacement_weight = 0;
    int gpu_device = 0;
    BioIKParams() { gpu_max_steps = 64; }
};

class BioIKKinematicsPlugin {
    const bio_ik::RobotModel* robot_model = nullptr;
    std::string group_name, base_frame;
    std::vector<std::string> joint_names, link_names, tip_frames_;
    mutable bio_ik::core::Engine engine;
    mutable std::vector<std::unique_ptr<bio_ik::Goal>> default_goals;
    double base_default[7];

public:
    // a batch that was submitted and not waited for yet (searchPositionIKBatchAsync)
    struct Pending {
        std::shared_ptr<bio_ik::core::Engine::Ticket> ticket;
    };

    BioIKKinematicsPlugin() {}
    BioIKKinematicsPlugin(const BioIKKinematicsPlugin&) = delete;

    // kinematics_plugin.cpp:362-374 (RobotModel overload) + load() :191-335.  Configuration errors throw, as the
    // reference's ERROR macro does; like the reference, a successful initialize returns true.
    bool initialize(const bio_ik::RobotModel& rm, const std::string& group, const std::string& base, const std::vector<std::string>& tip_frames,
                    double /*search_discretization*/ = 0.0, const BioIKParams& params = BioIKParams()) {
        robot_model = &rm;
        group_name = group, base_frame = base, tip_frames_ = tip_frames;
        const bio_ik::JointModelGroup& jmg = rm.groups.at(group);
        bio_ik::core::ModelView mv;
        mv.n_variables = rm.variable_names.size();
        mv.var_revolute.assign(mv.n_variables, 0);
        for (size_t l = 0; l < rm.joint_type.size(); l++) {
            if (rm.joint_type[l] == BIOIK_JOINT_REVOLUTE) mv.var_revolute[rm.joint_first_variable[l]] = 1;
            mv.has_mimic = mv.has_mimic || rm.joint_mimic[l] >= 0;
        }
        mv.var_bounded = rm.var_bounded, mv.var_min = rm.var_min, mv.var_max = rm.var_max;
        mv.var_max_velocity = rm.var_max_velocity;
        mv.var_prismatic.assign(mv.n_variables, 0);
        for (size_t l = 0; l < rm.joint_type.size(); l++)
            if (rm.joint_type[l] == BIOIK_JOINT_PRISMATIC) mv.var_prismatic[rm.joint_first_variable[l]] = 1;
        // (goals evaluated on the host -- JointFunctionGoal, LinkFunctionGoal, user subclasses: plugin_core.h, the hybrid path -- read link frames from here)
        mv.link_frame = [&rm](const std::string& link, const double* positions, double* frame7) {
            rm.linkTransform(rm.linkIndex(link), std::vector<double>(positions, positions + rm.variable_names.size()), frame7);
        };
        joint_names.clear();
        for (int j : jmg.active_joints) {
            joint_names.push_back(rm.joint_names[j]);
            const int nv = rm.joint_type[j] == BIOIK_JOINT_FLOATING ? 7 : (rm.joint_type[j] == BIOIK_JOINT_PLANAR ? 3 : 1);
            for (int vi = 0; vi < nv; vi++) mv.group_vars.push_back(rm.joint_first_variable[j] + vi);  // every variable of the joint (:473-484)
        }
        mv.group_joints = jmg.active_joints;
        mv.link_index = [&rm](const std::string& n) { return rm.linkIndex(n); };
        mv.variable_index = [&rm](const std::string& n) { return rm.variableIndex(n); };
        mv.joint_link_index = [&rm](const std::string& n) { return rm.jointIndex(n); };
        link_names = tip_frames;
        bio_ik::core::Settings s = params;
        s.devices = {params.gpu_device};
        engine.initialize(rm.desc(), mv, s);
        bio_ik::core::makeDefaultGoals(tip_frames, params.rotation_scale, params.position_only_ik, params.center_joints_weight, params.avoid_joint_limits_weight,
                                       params.minimal_displacement_weight, default_goals);  // :279-329
        rm.linkTransform(rm.linkIndex(base), rm.defaultPositions(), base_default);
        return true;
    }

    const std::vector<std::string>& getJointNames() const { return joint_names; }  // :130-133
    const std::vector<std::string>& getLinkNames() const { return link_names; }    // :135-138
    bool getPositionFK(const std::vector<std::string>&, const std::vector<double>&, std::vector<geometry_msgs::Pose>&) const { return false; }  // :140-145
    bool getPositionIK(const geometry_msgs::Pose&, const std::vector<double>&, std::vector<double>&, moveit_msgs::MoveItErrorCodes&,
                       const bio_ik::KinematicsQueryOptions& = bio_ik::KinematicsQueryOptions()) const { return false; }                        // :147-155
    bool supportsGroup(const bio_ik::JointModelGroup*, std::string* = nullptr) const { return true; }                                          // :657-662

    // The batched entry point without waiting: n independent queries sharing one goal structure are marshalled and enqueued; finish
    // with searchPositionIKBatchWait.  ik_poses [n][tips] (ignored with options.replace), ik_seed_states [n][group variables]
    // (referenced until the wait, like `options`, whose solution_fitness the wait writes).  Up to six batches per device may be in flight.
    Pending searchPositionIKBatchAsync(const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses, const std::vector<std::vector<double>>& ik_seed_states,
                                       const bio_ik::KinematicsQueryOptions& options = bio_ik::KinematicsQueryOptions(),
                                       const std::vector<double>* context_state = nullptr, double timeout = 0.0) const {
        bio_ik::core::Request rq;
        auto* bio = bio_ik::toBioIKKinematicsQueryOptions(&options);  // recognised by address, as the reference does (:75-101)
        const bool replace = bio && bio->replace;
        if (!replace)
            for (auto& g : default_goals) rq.goals.push_back(g.get());  // :550-552
        if (bio)
            for (auto& g : bio->goals) rq.goals.push_back(g.get());
        if (bio) rq.fixed_joints = bio->fixed_joints;
        rq.n_pose_goals = replace ? 0 : tip_frames_.size();
        rq.seed_states = &ik_seed_states;
        rq.context = context_state ? *context_state : robot_model->defaultPositions();
        if (context_state) robot_model->linkTransform(robot_model->linkIndex(base_frame), *context_state, rq.base_frame);
        else
            for (int c = 0; c < 7; c++) rq.base_frame[c] = base_default[c];
        for (size_t k = 0; k < ik_seed_states.size() && rq.n_pose_goals; k++)
            for (size_t t = 0; t < rq.n_pose_goals; t++) {
                const geometry_msgs::Pose& p = ik_poses.at(k).at(t);
                rq.tip_poses.insert(rq.tip_poses.end(), {p.position.x, p.position.y, p.position.z, p.orientation.x, p.orientation.y, p.orientation.z, p.orientation.w});
            }
        rq.timeout = timeout, rq.return_approximate_solution = options.return_approximate_solution, rq.bio = bio;
        return Pending{engine.submit(rq)};
    }
    // Returns true iff every query produced an acceptable solution; per-query verdicts in error_codes.
    bool searchPositionIKBatchWait(Pending& pending, std::vector<std::vector<double>>& solutions, std::vector<moveit_msgs::MoveItErrorCodes>& error_codes) const {
        std::vector<uint8_t> ok;
        const bool all_ok = engine.wait(*pending.ticket, solutions, ok);
        error_codes.assign(ok.size(), moveit_msgs::MoveItErrorCodes());
        for (size_t k = 0; k < ok.size(); k++) error_codes[k].val = ok[k] ? moveit_msgs::MoveItErrorCodes::SUCCESS : moveit_msgs::MoveItErrorCodes::NO_IK_SOLUTION;
        return all_ok;
    }
    bool searchPositionIKBatch(const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses, const std::vector<std::vector<double>>& ik_seed_states,
                               std::vector<std::vector<double>>& solutions, std::vector<moveit_msgs::MoveItErrorCodes>& error_codes,
                               const bio_ik::KinematicsQueryOptions& options = bio_ik::KinematicsQueryOptions(),
                               const std::vector<double>* context_state = nullptr, double timeout = 0.0) const {
        Pending p = searchPositionIKBatchAsync(ik_poses, ik_seed_states, options, context_state, timeout);
        return searchPositionIKBatchWait(p, solutions, error_codes);
    }

    // kinematics_plugin.cpp:437-655 (the multi-pose overload every other overload forwards to)
    bool searchPositionIK(const std::vector<geometry_msgs::Pose>& ik_poses, const std::vector<double>& ik_seed_state, double timeout,
                          const std::vector<double>& /*consistency_limits*/, std::vector<double>& solution, const IKCallbackFn& solution_callback,
                          moveit_msgs::MoveItErrorCodes& error_code, const bio_ik::KinematicsQueryOptions& options = bio_ik::KinematicsQueryOptions(),
                          const std::vector<double>* context_state = nullptr) const {
        std::vector<std::vector<double>> sols;
        std::vector<moveit_msgs::MoveItErrorCodes> codes;
        bool ok = searchPositionIKBatch({ik_poses}, {ik_seed_state}, sols, codes, options, context_state, timeout);
        solution = sols.empty() || sols[0].empty() ? ik_seed_state : sols[0];
        if (!ok) {
            error_code.val = moveit_msgs::MoveItErrorCodes::NO_IK_SOLUTION;
            return false;
        }
        if (solution_callback) {  // :644-649
            solution_callback(ik_poses.empty() ? geometry_msgs::Pose() : ik_poses.front(), solution, error_code);
            return error_code.val == moveit_msgs::MoveItErrorCodes::SUCCESS;
        }
        error_code.val = moveit_msgs::MoveItErrorCodes::SUCCESS;
        return true;
    }
    // single-pose overloads (:376-430)
    bool searchPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, double timeout, std::vector<double>& solution,
                          moveit_msgs::MoveItErrorCodes& error_code, const bio_ik::KinematicsQueryOptions& options = bio_ik::KinematicsQueryOptions()) const {
        return searchPositionIK(std::vector<geometry_msgs::Pose>{ik_pose}, ik_seed_state, timeout, std::vector<double>(), solution, IKCallbackFn(), error_code,
                                options);
    }
    bool searchPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, double timeout, std::vector<double>& solution,
                          const IKCallbackFn& solution_callback, moveit_msgs::MoveItErrorCodes& error_code,
                          const bio_ik::KinematicsQueryOptions& options = bio_ik::KinematicsQueryOptions()) const {
        return searchPositionIK(std::vector<geometry_msgs::Pose>{ik_pose}, ik_seed_state, timeout, std::vector<double>(), solution, solution_callback,
                                error_code, options);
    }
};

}  // namespace bio_ik_kinematics_plugin
