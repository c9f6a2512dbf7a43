// bio_ik/kinematics_plugin.h — `bio_ik_kinematics_plugin::BioIKKinematicsPlugin` over the HIP C-ABI.
//
// C++ host-side mirror of the reference plugin (src/kinematics_plugin.cpp:117-671): same method names, argument
// meaning and error behaviour as its kinematics::KinematicsBase implementation, with light stand-ins for the ROS
// message types (geometry_msgs::Pose, moveit_msgs::MoveItErrorCodes) because ROS / MoveIt are not installed here.
// What the reference does between `problem.initialize(...)` and `ik->getSolution()` (:560-578) is one call of
// `bioik_solve_batch`; everything around it (seed -> state :465-485, goal frames into the model frame :487-502,
// default goals :279-329, angle wrapping :580-616, result mapping :619-629, error codes :632-654) is restated here.
// `searchPositionIKBatch` is the additive batched entry point.  Header-only; link with libbioik_hip.so.
#pragma once
#include <cfloat>
#include <functional>
#include <map>
#include <memory>
#include <sstream>

#include "bio_ik.h"
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

struct BioIKParams {  // the kinematics.yaml keys of the reference (kinematics_plugin.cpp:243-328) + additive gpu_* keys
    std::string mode = "bio2_memetic";
    int random_seed = 0;
    double dpos = DBL_MAX, drot = DBL_MAX, dtwist = 1e-5;
    bool no_wipeout = false;
    double rotation_scale = 0.5;
    bool position_only_ik = false;
    double center_joints_weight = 0, avoid_joint_limits_weight = 0, minimal_displacement_weight = 0;
    int gpu_population = 128, gpu_islands = 1, gpu_max_steps = 64, gpu_device = 0;
    std::string gpu_fk = "exact";
};

class BioIKKinematicsPlugin {
    const bio_ik::RobotModel* robot_model = nullptr;
    std::string group_name, base_frame;
    std::vector<std::string> joint_names, link_names, tip_frames_;
    std::vector<int> group_vars;
    BioIKParams ikparams;
    bioik_model* model = nullptr;
    mutable std::vector<std::unique_ptr<bio_ik::Goal>> default_goals;
    mutable std::map<std::string, bioik_problem*> problems;  // one compiled problem per goal structure
    double base_default[7];

    static int modeOf(const std::string& name) {
        if (name == "bio2") return BIOIK_MODE_BIO2;
        if (name == "bio2_memetic") return BIOIK_MODE_BIO2_MEMETIC;
        if (name == "bio2_memetic_l") return BIOIK_MODE_BIO2_MEMETIC_L;
        if (name == "gd_c") return BIOIK_MODE_GD_C;  // src/ik_gradient.cpp:263
        if (name == "gd") return BIOIK_MODE_GD;      // :253
        if (name == "jac") return BIOIK_MODE_JAC;    // src/ik_gradient.cpp:289
        throw std::runtime_error("unknown solver mode " + name);  // IKFactory::create -> ERROR, src/utils.h:436
    }
    bioik_problem* problemFor(const std::vector<const bio_ik::Goal*>& goals, const std::vector<std::string>& fixed) const {
        std::ostringstream key;
        key << std::hexfloat;  // exact weight bits: goals whose weights differ in any digit get their own compiled problem
        for (auto* g : goals) key << g->gpuOpcode() << ':' << g->gpuLinkName() << ':' << g->gpuVariableName() << ':' << g->getWeight() << ':' << g->isSecondary() << ';';
        for (auto& f : fixed) key << '#' << f;
        auto it = problems.find(key.str());
        if (it != problems.end()) return it->second;
        std::vector<bioik_goal_desc> gd;
        for (auto* g : goals) {
            if (g->gpuOpcode() < 0) throw std::runtime_error("goal has no device implementation (host-callback goal)");
            bioik_goal_desc d{g->gpuOpcode(), -1, -1, g->isSecondary() ? 1 : 0, g->getWeight()};
            if (!g->gpuLinkName().empty()) d.link = robot_model->linkIndex(g->gpuLinkName());
            if (!g->gpuVariableName().empty()) d.variable = robot_model->variableIndex(g->gpuVariableName());
            gd.push_back(d);
        }
        std::vector<int32_t> fixed_idx;
        for (auto& f : fixed) fixed_idx.push_back(robot_model->jointIndex(f));
        const bio_ik::JointModelGroup& jmg = robot_model->groups.at(group_name);
        bioik_problem_desc pd{};
        pd.struct_size = sizeof(pd);
        pd.n_group_joints = (uint32_t)jmg.active_joints.size(), pd.group_joints = jmg.active_joints.data();
        pd.n_goals = (uint32_t)gd.size(), pd.goals = gd.data();
        pd.n_fixed_joints = (uint32_t)fixed_idx.size(), pd.fixed_joints = fixed_idx.data();
        bioik_problem* p = nullptr;
        if (bioik_problem_create(model, &pd, &p) != BIOIK_OK) throw std::runtime_error(bioik_last_error());
        problems[key.str()] = p;
        return p;
    }

public:
    BioIKKinematicsPlugin() {}
    BioIKKinematicsPlugin(const BioIKKinematicsPlugin&) = delete;
    ~BioIKKinematicsPlugin() {
        for (auto& kv : problems) bioik_problem_destroy(kv.second);
        bioik_model_destroy(model);
    }

    // kinematics_plugin.cpp:362-374 (RobotModel overload) + load() :191-335.  Configuration errors throw, as the
    // reference's ERROR macro does; like the reference, a successful initialize returns true.
    bool initialize(const bio_ik::RobotModel& rm, const std::string& group, const std::string& base, const std::vector<std::string>& tip_frames,
                    double /*search_discretization*/ = 0.0, const BioIKParams& params = BioIKParams()) {
        robot_model = &rm;
        group_name = group, base_frame = base, tip_frames_ = tip_frames, ikparams = params;
        modeOf(params.mode);
        const bio_ik::JointModelGroup& jmg = rm.groups.at(group);
        joint_names.clear(), group_vars.clear();
        for (int j : jmg.active_joints) {
            joint_names.push_back(rm.joint_names[j]);
            const int nv = rm.joint_type[j] == BIOIK_JOINT_FLOATING ? 7 : (rm.joint_type[j] == BIOIK_JOINT_PLANAR ? 3 : 1);
            for (int vi = 0; vi < nv; vi++) group_vars.push_back(rm.joint_first_variable[j] + vi);  // every variable of the joint (:473-484)
        }
        link_names = tip_frames;
        bioik_model_desc md = rm.desc();
        if (bioik_model_create(&md, params.gpu_device, &model) != BIOIK_OK) throw std::runtime_error(bioik_last_error());
        default_goals.clear();  // :279-329
        for (auto& tip : tip_frames) {
            auto* g = new bio_ik::PoseGoal();
            g->setLinkName(tip);
            g->setRotationScale(params.position_only_ik ? 0.0 : params.rotation_scale);
            default_goals.emplace_back(g);
        }
        if (params.center_joints_weight > 0) default_goals.emplace_back(new bio_ik::CenterJointsGoal(params.center_joints_weight));
        if (params.avoid_joint_limits_weight > 0) default_goals.emplace_back(new bio_ik::AvoidJointLimitsGoal(params.avoid_joint_limits_weight));
        if (params.minimal_displacement_weight > 0) default_goals.emplace_back(new bio_ik::MinimalDisplacementGoal(params.minimal_displacement_weight));
        rm.linkTransform(rm.linkIndex(base), rm.defaultPositions(), base_default);
        return true;
    }

    const std::vector<std::string>& getJointNames() const { return joint_names; }  // :130-133
    const std::vector<std::string>& getLinkNames() const { return link_names; }    // :135-138
    bool getPositionFK(const std::vector<std::string>&, const std::vector<double>&, std::vector<geometry_msgs::Pose>&) const { return false; }  // :140-145
    bool getPositionIK(const geometry_msgs::Pose&, const std::vector<double>&, std::vector<double>&, moveit_msgs::MoveItErrorCodes&,
                       const bio_ik::KinematicsQueryOptions& = bio_ik::KinematicsQueryOptions()) const { return false; }                        // :147-155
    bool supportsGroup(const bio_ik::JointModelGroup*, std::string* = nullptr) const { return true; }                                          // :657-662

    // The additive batched entry point: n independent queries sharing one goal structure, one bioik_solve_batch call.
    // ik_poses [n][tips] (ignored with options.replace), ik_seed_states [n][group variables].  Returns true iff every
    // query produced an acceptable solution; per-query verdicts in error_codes.
    bool searchPositionIKBatch(const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses, const std::vector<std::vector<double>>& ik_seed_states,
                               std::vector<std::vector<double>>& solutions, std::vector<moveit_msgs::MoveItErrorCodes>& error_codes,
                               const bio_ik::KinematicsQueryOptions& options = bio_ik::KinematicsQueryOptions(),
                               const std::vector<double>* context_state = nullptr, double timeout = 0.0) const {
        const size_t n = ik_seed_states.size(), V = robot_model->variable_names.size();
        auto* bio = bio_ik::toBioIKKinematicsQueryOptions(&options);  // recognised by address, as the reference does (:75-101)
        std::vector<const bio_ik::Goal*> all_goals;
        if (!bio || !bio->replace)
            for (auto& g : default_goals) all_goals.push_back(g.get());  // :550-552
        if (bio)
            for (auto& g : bio->goals) all_goals.push_back(g.get());
        bioik_problem* problem = problemFor(all_goals, bio ? bio->fixed_joints : std::vector<std::string>());
        const size_t P = (size_t)bioik_problem_param_count(problem);
        // seed -> full state (:465-485)
        std::vector<double> state(n * V), params(n * P), sol(n * V), fit(n);
        std::vector<int32_t> suc(n), steps(n);
        std::vector<double> base = context_state ? *context_state : robot_model->defaultPositions();
        for (size_t k = 0; k < n; k++) {
            for (size_t v = 0; v < V; v++) state[k * V + v] = base[v];
            for (size_t i = 0; i < group_vars.size(); i++) state[k * V + group_vars[i]] = ik_seed_states[k].at(i);
        }
        // per-query goal numbers; default pose goals move into the model frame (:487-502, :540-546)
        double r[7];
        if (context_state) robot_model->linkTransform(robot_model->linkIndex(base_frame), *context_state, r);
        for (size_t k = 0; k < n; k++) {
            std::vector<double> row;
            size_t gi = 0;
            for (auto* g : all_goals) {
                if ((!bio || !bio->replace) && gi < tip_frames_.size()) {
                    const geometry_msgs::Pose& p = ik_poses.at(k).at(gi);
                    double pose[7] = {p.position.x, p.position.y, p.position.z, p.orientation.x, p.orientation.y, p.orientation.z, p.orientation.w};
                    double m[7];
                    bio_ik::RobotModel::concat(context_state ? r : base_default, pose, m);
                    auto* pg = static_cast<bio_ik::PoseGoal*>(const_cast<bio_ik::Goal*>(g));
                    pg->setPosition(bio_ik::Vector3(m[0], m[1], m[2]));
                    pg->setOrientation(bio_ik::Quaternion(m[3], m[4], m[5], m[6]));
                }
                g->gpuParams(row);
                gi++;
            }
            for (size_t i = 0; i < P; i++) params[k * P + i] = row.at(i);
        }
        bioik_solve_params sp;
        bioik_default_solve_params(&sp);
        sp.mode = modeOf(ikparams.mode);
        sp.fk_mode = ikparams.gpu_fk == "linear" ? BIOIK_FK_LINEAR : BIOIK_FK_EXACT;
        sp.population = ikparams.gpu_population, sp.islands = ikparams.gpu_islands, sp.max_steps = ikparams.gpu_max_steps;
        sp.random_seed = (uint64_t)ikparams.random_seed;
        sp.dpos = ikparams.dpos, sp.drot = ikparams.drot, sp.dtwist = ikparams.dtwist;
        sp.no_wipeout = ikparams.no_wipeout;
        sp.timeout = timeout > 0.0 ? timeout : 0.0;  // ik_parallel.h:160: wall-clock budget of the call, honoured on the device
        solutions.assign(n, std::vector<double>());
        error_codes.assign(n, moveit_msgs::MoveItErrorCodes());
        if (bioik_solve_batch(problem, &sp, n, state.data(), params.data(), sol.data(), fit.data(), suc.data(), steps.data()) != BIOIK_OK) {
            for (auto& e : error_codes) e.val = moveit_msgs::MoveItErrorCodes::NO_IK_SOLUTION;  // HIP errors never abort the caller
            return false;
        }
        std::vector<int32_t> active(bioik_problem_active_variable_count(problem));
        bioik_problem_active_variables(problem, active.data());
        bool all_ok = true;
        for (size_t k = 0; k < n; k++) {
            double* st = &sol[k * V];
            // wrap angles (:580-613); skipped, as in the reference, for models with mimic joints (:583-584)
            bool has_mimic = false;
            for (int m : robot_model->joint_mimic) has_mimic = has_mimic || m >= 0;
            for (int ivar : active) {
                double v = st[ivar];
                bool revolute = false;
                for (size_t l = 0; l < robot_model->joint_type.size(); l++)
                    if (robot_model->joint_first_variable[l] == ivar) revolute = robot_model->joint_type[l] == BIOIK_JOINT_REVOLUTE;
                if (revolute && !has_mimic) {
                    double rr = state[k * V + ivar], lo = robot_model->var_min[ivar], hi = robot_model->var_max[ivar];
                    if (rr < v - M_PI || rr > v + M_PI) {
                        v -= rr, v /= (2 * M_PI), v += 0.5, v -= std::floor(v), v -= 0.5, v *= (2 * M_PI), v += rr;
                    }
                    if (v > hi) v -= std::ceil(std::max(0.0, v - hi) / (2 * M_PI)) * (2 * M_PI);
                    if (v < lo) v += std::ceil(std::max(0.0, lo - v) / (2 * M_PI)) * (2 * M_PI);
                    if (v < lo) v = lo;
                    if (v > hi) v = hi;
                }
                st[ivar] = v;
            }
            for (size_t v = 0; v < V; v++) {  // RobotModel::enforcePositionBounds (:616): clamp bounded variables, wrap continuous joints
                if (robot_model->var_bounded[v]) {
                    st[v] = std::min(std::max(st[v], robot_model->var_min[v]), robot_model->var_max[v]);
                } else if (st[v] <= -M_PI || st[v] > M_PI) {
                    bool revolute = false;
                    for (size_t l = 0; l < robot_model->joint_type.size(); l++)
                        if (robot_model->joint_first_variable[l] == (int)v) revolute = robot_model->joint_type[l] == BIOIK_JOINT_REVOLUTE;
                    if (revolute) {
                        st[v] = std::fmod(st[v], 2 * M_PI);
                        if (st[v] <= -M_PI) st[v] += 2 * M_PI;
                        else if (st[v] > M_PI) st[v] -= 2 * M_PI;
                    }
                }
            }
            for (int gv : group_vars) solutions[k].push_back(st[gv]);  // :619-629
            bool ok = suc[k] || options.return_approximate_solution;  // :638-641
            error_codes[k].val = ok ? moveit_msgs::MoveItErrorCodes::SUCCESS : moveit_msgs::MoveItErrorCodes::NO_IK_SOLUTION;
            all_ok = all_ok && ok;
        }
        if (bio && n) bio->solution_fitness = fit[n - 1];  // :632-634
        return all_ok;
    }

    // kinematics_plugin.cpp:437-655 (the multi-pose overload every other overload forwards to)
    bool searchPositionIK(const std::vector<geometry_msgs::Pose>& ik_poses, const std::vector<double>& ik_seed_state, double timeout,
                          const std::vector<double>& /*consistency_limits*/, std::vector<double>& solution, const IKCallbackFn& solution_callback,
                          moveit_msgs::MoveItErrorCodes& error_code, const bio_ik::KinematicsQueryOptions& options = bio_ik::KinematicsQueryOptions(),
                          const std::vector<double>* context_state = nullptr) const {
        std::vector<std::vector<double>> sols;
        std::vector<moveit_msgs::MoveItErrorCodes> codes;
        bool ok = searchPositionIKBatch({ik_poses}, {ik_seed_state}, sols, codes, options, context_state, timeout);
        solution = sols.empty() ? ik_seed_state : sols[0];
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
