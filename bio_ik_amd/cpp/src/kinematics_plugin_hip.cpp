// kinematics_plugin_hip.cpp — `bio_ik_kinematics_plugin::BioIKKinematicsPlugin : kinematics::KinematicsBase` over the HIP C-ABI.
//
// The translation unit MoveIt loads through pluginlib as `bio_ik/BioIKKinematicsPlugin` from `libbio_ik`
// (bio_ik_kinematics_description.xml), replacing the reference's src/kinematics_plugin.cpp:117-671: the same virtuals with the same
// argument meaning, parameter names (kinematics.yaml keys, :243-328) and error behaviour.  Where the reference runs
// `problem.initialize(...); ik->initialize(problem); ik->solve();` on CPU threads (:560-578), this plugin marshals
// moveit::core::RobotModel into the flat `bioik_model_desc` once, compiles one `bioik_problem` per goal structure, and calls
// `bioik_solve_batch` of libbioik_hip.so (include/bioik_hip.h) — the MI355X kernels.  There is no CPU solver behind it: without a
// HIP device `initialize` throws, as the reference's ERROR macro does for configuration errors (src/utils.h:122-129).
//
// Built against the real MoveIt / ROS / Eigen / pluginlib headers in a ROS workspace (CMakeLists.txt), and against the minimal
// stand-ins of ../standin in this repository's build image, where none of them is installed (Makefile).
#include <cfloat>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <stdexcept>

#include <Eigen/Geometry>
#include <geometry_msgs/Pose.h>
#include <moveit/kinematics_base/kinematics_base.h>
#include <moveit/robot_model/robot_model.h>
#include <moveit/robot_state/robot_state.h>
#include <moveit_msgs/MoveItErrorCodes.h>
#include <pluginlib/class_list_macros.h>
#include <ros/ros.h>

#define BIOIK_WITH_KINEMATICS_BASE 1
#include <bio_ik/bio_ik.h>
#include <bio_ik/kinematics_plugin_hip.h>
#include <bio_ik/plugin_core.h>
#include <moveit/rdf_loader/rdf_loader.h>

namespace bio_ik_kinematics_plugin {

// Fallback for MoveIt versions without KinematicsBase::lookupParam (reference :108-115): the plugin's private namespace
template <class T>
static void lookupParam(const std::string& param, T& val, const T& default_val) {
    ros::NodeHandle nodeHandle("~");
    val = nodeHandle.param(param, default_val);
}

namespace {
struct Frame7 {  // px py pz qx qy qz qw
    double v[7];
};
Frame7 toFrame(const Eigen::Isometry3d& T) {
    const Eigen::Quaterniond q(T.rotation());
    Frame7 f;
    f.v[0] = T.translation().x(), f.v[1] = T.translation().y(), f.v[2] = T.translation().z();
    f.v[3] = q.x(), f.v[4] = q.y(), f.v[5] = q.z(), f.v[6] = q.w();
    return f;
}
Frame7 toFrame(const geometry_msgs::Pose& p) {
    Frame7 f;
    f.v[0] = p.position.x, f.v[1] = p.position.y, f.v[2] = p.position.z;
    f.v[3] = p.orientation.x, f.v[4] = p.orientation.y, f.v[5] = p.orientation.z, f.v[6] = p.orientation.w;
    return f;
}
}  // namespace

// What RobotFK / RobotInfo read from moveit::core::RobotModel (src/forward_kinematics.h:192-213, 230-246, 268-329;
// include/bio_ik/robot_info.h:70-106), flattened for bioik_model_create.  Link i carries its parent joint.
struct FlatModel {
    std::vector<int32_t> link_parent, joint_type, joint_first_variable, joint_mimic;
    std::vector<double> link_origin, joint_axis, joint_mimic_factor, joint_mimic_offset, var_min, var_max, var_max_velocity, link_mass, link_center;
    std::vector<uint8_t> var_bounded;
    explicit FlatModel(const moveit::core::RobotModel& rm) {
        const auto& links = rm.getLinkModels();
        for (const moveit::core::LinkModel* l : links) {
            const moveit::core::JointModel* j = l->getParentJointModel();
            link_parent.push_back(l->getParentLinkModel() ? (int32_t)l->getParentLinkModel()->getLinkIndex() : -1);
            const Frame7 o = toFrame(l->getJointOriginTransform());
            link_origin.insert(link_origin.end(), o.v, o.v + 7);
            int t = BIOIK_JOINT_FIXED;
            double ax[3] = {0, 0, 0};
            switch (j->getType()) {
                case moveit::core::JointModel::REVOLUTE: {
                    t = BIOIK_JOINT_REVOLUTE;
                    const auto& a = static_cast<const moveit::core::RevoluteJointModel*>(j)->getAxis();
                    ax[0] = a.x(), ax[1] = a.y(), ax[2] = a.z();
                    break;
                }
                case moveit::core::JointModel::PRISMATIC: {
                    t = BIOIK_JOINT_PRISMATIC;
                    const auto& a = static_cast<const moveit::core::PrismaticJointModel*>(j)->getAxis();
                    ax[0] = a.x(), ax[1] = a.y(), ax[2] = a.z();
                    break;
                }
                case moveit::core::JointModel::FLOATING: t = BIOIK_JOINT_FLOATING; break;
                case moveit::core::JointModel::PLANAR: t = BIOIK_JOINT_PLANAR; break;
                default: break;
            }
            joint_type.push_back(t);
            joint_axis.insert(joint_axis.end(), ax, ax + 3);
            joint_first_variable.push_back(j->getVariableCount() ? (int32_t)j->getFirstVariableIndex() : -1);
            joint_mimic.push_back(j->getMimic() ? (int32_t)j->getMimic()->getChildLinkModel()->getLinkIndex() : -1);
            joint_mimic_factor.push_back(j->getMimicFactor());
            joint_mimic_offset.push_back(j->getMimicOffset());
            // urdf <inertial> of the link, what BalanceGoal::describe reads (src/goal_types.cpp:236-247)
            double mass = 0.0, c[3] = {0, 0, 0};
            if (rm.getURDF()) {
                auto link_urdf = rm.getURDF()->getLink(l->getName());
                if (link_urdf && link_urdf->inertial) {
                    mass = link_urdf->inertial->mass;
                    c[0] = link_urdf->inertial->origin.position.x, c[1] = link_urdf->inertial->origin.position.y, c[2] = link_urdf->inertial->origin.position.z;
                }
            }
            link_mass.push_back(mass);
            link_center.insert(link_center.end(), c, c + 3);
        }
        for (const std::string& name : rm.getVariableNames()) {
            const moveit::core::VariableBounds& b = rm.getVariableBounds(name);
            var_min.push_back(b.min_position_), var_max.push_back(b.max_position_);
            var_bounded.push_back(b.position_bounded_ ? 1 : 0);
            var_max_velocity.push_back(b.max_velocity_);
        }
    }
    bioik_model_desc desc() const {
        bioik_model_desc d{};
        d.struct_size = sizeof(d);
        d.n_links = (uint32_t)link_parent.size(), d.n_variables = (uint32_t)var_min.size();
        d.link_parent = link_parent.data(), d.link_origin = link_origin.data(), d.joint_type = joint_type.data(), d.joint_axis = joint_axis.data();
        d.joint_first_variable = joint_first_variable.data(), d.joint_mimic = joint_mimic.data();
        d.joint_mimic_factor = joint_mimic_factor.data(), d.joint_mimic_offset = joint_mimic_offset.data();
        d.var_min = var_min.data(), d.var_max = var_max.data(), d.var_bounded = var_bounded.data(), d.var_max_velocity = var_max_velocity.data();
        d.link_mass = link_mass.data(), d.link_center = link_center.data();
        return d;
    }
};

struct BioIKKinematicsPlugin : kinematics::KinematicsBase {
    std::vector<std::string> joint_names, link_names;
    moveit::core::RobotModelConstPtr robot_model;
    const moveit::core::JointModelGroup* joint_model_group = nullptr;
    mutable std::vector<double> state;
    mutable std::unique_ptr<moveit::core::RobotState> temp_state, host_state;
    mutable std::vector<std::unique_ptr<bio_ik::Goal>> default_goals;
    mutable std::mutex mutex;  // the reference's instance is not re-entrant either (mutable members, :121-124); here calls serialise

    // IKParams (src/utils.h:64-85) restricted to what the device path reads, + the additive gpu_* keys (bio_ik/plugin_core.h)
    bio_ik::core::Settings settings;
    std::unique_ptr<FlatModel> flat;
    // everything between the arguments of searchPositionIK and the C-ABI: the implementation shared with the other faces of the plugin
    mutable bio_ik::core::Engine engine;

    BioIKKinematicsPlugin() {}
    ~BioIKKinematicsPlugin() override {}

    const std::vector<std::string>& getJointNames() const override { return joint_names; }  // :130-133
    const std::vector<std::string>& getLinkNames() const override { return link_names; }    // :135-138
    bool getPositionFK(const std::vector<std::string>&, const std::vector<double>&, std::vector<geometry_msgs::Pose>&) const override { return false; }  // :140-145
    bool getPositionIK(const geometry_msgs::Pose&, const std::vector<double>&, std::vector<double>&, moveit_msgs::MoveItErrorCodes&,
                       const kinematics::KinematicsQueryOptions& = kinematics::KinematicsQueryOptions()) const override {
        return false;  // :147-155
    }

    // :167-189: the model behind a robot description on the parameter server (URDF + SRDF through rdf_loader), built once per name
    static moveit::core::RobotModelConstPtr modelOfDescription(const std::string& robot_description) {
        static std::mutex mutex;
        static std::map<std::string, moveit::core::RobotModelConstPtr> models;
        std::lock_guard<std::mutex> lock(mutex);
        auto it = models.find(robot_description);
        if (it != models.end()) return it->second;
        rdf_loader::RDFLoader loader(robot_description);
        if (!loader.getURDF() || !loader.getSRDF()) return nullptr;  // "URDF and SRDF must be loaded for kinematics solver to work." (:178-181)
        moveit::core::RobotModelConstPtr model(new moveit::core::RobotModel(loader.getURDF(), loader.getSRDF()));
        models[robot_description] = model;
        return model;
    }

    // :191-335
    bool load(const moveit::core::RobotModelConstPtr& model_arg, const std::string& robot_description, const std::string& group_name) {
        engine.release();  // (a re-initialisation starts from nothing, also when it fails below)
        default_goals.clear();
        const moveit::core::RobotModelConstPtr model_ptr = model_arg ? model_arg : modelOfDescription(robot_description);  // :205-208
        if (!model_ptr) {
            robot_model.reset();
            return false;  // "failed to load robot model" (:209-212)
        }
        robot_model = model_ptr;
        joint_model_group = robot_model->getJointModelGroup(group_name);
        if (!joint_model_group) {  // "failed to get joint model group" (:215-218)
            robot_model.reset();
            return false;
        }
        joint_names.clear();
        for (auto* joint_model : joint_model_group->getJointModels())
            if (joint_model->getName() != base_frame_ && joint_model->getType() != moveit::core::JointModel::UNKNOWN &&
                joint_model->getType() != moveit::core::JointModel::FIXED)
                joint_names.push_back(joint_model->getName());
        auto tips2 = tip_frames_;
        joint_model_group->getEndEffectorTips(tips2);
        if (!tips2.empty()) tip_frames_ = tips2;
        link_names = tip_frames_;

        bio_ik::core::Settings& p = settings;
        p = bio_ik::core::Settings();
        lookupParam("mode", p.mode, std::string("bio2_memetic"));
        lookupParam("random_seed", p.random_seed, static_cast<int>(std::random_device()()));
        lookupParam("dpos", p.dpos, DBL_MAX);
        lookupParam("drot", p.drot, DBL_MAX);
        lookupParam("dtwist", p.dtwist, 1e-5);
        lookupParam("no_wipeout", p.no_wipeout, false);
        lookupParam("gpu_population", p.gpu_population, 128);   // children per species and generation (reference: 16, ik_evolution_2.cpp:138)
        lookupParam("gpu_islands", p.gpu_islands, 0);  // 0: as many as the idle part of the chip carries (BIOIK_ISLANDS_AUTO)
        lookupParam("gpu_island_sync", p.gpu_island_sync, true);  // islands stop once one of them has a solution, as the reference's solver threads do
        lookupParam("gpu_host_goal_candidates", p.gpu_host_goal_candidates, 4);  // candidates per query that the host scores when the goal list holds callback goals
        lookupParam("gpu_max_steps", p.gpu_max_steps, 4096);    // safety cap; the caller's timeout is what normally ends a query
        lookupParam("gpu_fk", p.gpu_fk, std::string("exact"));  // "exact" | "linear" (the reference's linearised phenotypes)
        lookupParam("gpu_schedule", p.gpu_schedule, std::string("auto"));  // "auto" | "latency" | "throughput" (plugin_core.h: Settings)
        lookupParam("gpu_reproducible_calls", p.gpu_reproducible_calls, false);  // true: a repeated call replays the same random streams
        int gpu_device = 0;
        lookupParam("gpu_device", gpu_device, 0);
        std::string gpu_devices;  // kinematics.yaml `gpu_devices: "0,1,2,3"` (default: the single `gpu_device`): a batch is sharded over them
        lookupParam("gpu_devices", gpu_devices, std::string());
        p.devices.clear();
        {
            std::stringstream ss(gpu_devices);
            for (std::string item; std::getline(ss, item, ',');)
                if (!item.empty()) p.devices.push_back(std::stoi(item));
        }
        if (p.devices.empty()) p.devices.push_back(gpu_device);

        temp_state.reset(new moveit::core::RobotState(robot_model));
        flat.reset(new FlatModel(*robot_model));
        // what the shared implementation needs to know about the model (bio_ik/plugin_core.h: ModelView)
        bio_ik::core::ModelView mv;
        const moveit::core::RobotModel* rm = robot_model.get();
        mv.n_variables = rm->getVariableCount();
        mv.var_revolute.assign(mv.n_variables, 0);
        for (size_t v = 0; v < mv.n_variables; v++) {
            const moveit::core::JointModel* jm = rm->getJointOfVariable((int)v);
            mv.var_revolute[v] = jm->getType() == moveit::core::JointModel::REVOLUTE ? 1 : 0;
            const moveit::core::VariableBounds& b = jm->getVariableBounds()[v - (size_t)jm->getFirstVariableIndex()];
            mv.var_bounded.push_back(b.position_bounded_ ? 1 : 0);
            mv.var_min.push_back(b.min_position_), mv.var_max.push_back(b.max_position_);
            mv.var_max_velocity.push_back(b.max_velocity_);
            mv.var_prismatic.push_back(jm->getType() == moveit::core::JointModel::PRISMATIC ? 1 : 0);
        }
        // goals that are evaluated on the host (JointFunctionGoal, LinkFunctionGoal, user subclasses: bio_ik/plugin_core.h, the hybrid path) read link
        // frames through MoveIt's own forward kinematics
        host_state.reset(new moveit::core::RobotState(robot_model));
        moveit::core::RobotState* hs = host_state.get();
        const size_t n_var = mv.n_variables;
        mv.link_frame = [hs, n_var](const std::string& link, const double* positions, double* frame7) {
            hs->setVariablePositions(std::vector<double>(positions, positions + n_var));
            hs->update();
            const Frame7 f = toFrame(hs->getGlobalLinkTransform(link));
            for (int c = 0; c < 7; c++) frame7[c] = f.v[c];
        };
        mv.has_mimic = !rm->getMimicJointModels().empty();
        for (auto& joint_name : joint_names) {  // seed / solution vectors: the variables of the group's joints in this order (:473-484, :619-629)
            auto* joint_model = rm->getJointModel(joint_name);
            if (!joint_model) continue;
            for (size_t vi = 0; vi < joint_model->getVariableCount(); vi++) mv.group_vars.push_back((int)(joint_model->getFirstVariableIndex() + vi));
        }
        for (auto* j : joint_model_group->getActiveJointModels()) mv.group_joints.push_back((int32_t)j->getChildLinkModel()->getLinkIndex());
        mv.link_index = [rm](const std::string& n) {
            auto* l = rm->getLinkModel(n);
            return l ? (int)l->getLinkIndex() : -1;
        };
        mv.variable_index = [rm](const std::string& n) { return (int)rm->getVariableIndex(n); };
        mv.joint_link_index = [rm](const std::string& n) {
            auto* j = rm->getJointModel(n);
            return j ? (int)j->getChildLinkModel()->getLinkIndex() : -1;
        };
        mv.enforce_bounds = [rm](double* st) { rm->enforcePositionBounds(st); };
        engine.initialize(flat->desc(), mv, p);  // (throws on an unknown solver mode or a missing HIP device: configuration errors)

        double rotation_scale = 0.5, center = 0, avoid = 0, minimal = 0;  // :279-329
        bool position_only_ik = false;
        lookupParam("rotation_scale", rotation_scale, rotation_scale);
        lookupParam("position_only_ik", position_only_ik, position_only_ik);
        lookupParam("center_joints_weight", center, 0.0);
        lookupParam("avoid_joint_limits_weight", avoid, 0.0);
        lookupParam("minimal_displacement_weight", minimal, 0.0);
        bio_ik::core::makeDefaultGoals(tip_frames_, rotation_scale, position_only_ik, center, avoid, minimal, default_goals);
        return true;
    }

    // :337-374.  Like the reference, initialize returns true even when load() reported a failure.
    bool initialize(const std::string& robot_description, const std::string& group_name, const std::string& base_frame, const std::string& tip_frame,
                    double search_discretization) override {
        std::vector<std::string> tip_frames;
        tip_frames.push_back(tip_frame);
        initialize(robot_description, group_name, base_frame, tip_frames, search_discretization);
        return true;
    }
    bool initialize(const std::string& robot_description, const std::string& group_name, const std::string& base_frame, const std::vector<std::string>& tip_frames,
                    double search_discretization) override {
        setValues(robot_description, group_name, base_frame, tip_frames, search_discretization);
        load(moveit::core::RobotModelConstPtr(), robot_description, group_name);
        return true;
    }
    bool initialize(const moveit::core::RobotModel& rm, const std::string& group_name, const std::string& base_frame, const std::vector<std::string>& tip_frames,
                    double search_discretization) override {
        setValues("", group_name, base_frame, tip_frames, search_discretization);
        load(moveit::core::RobotModelConstPtr(&rm, [](const moveit::core::RobotModel*) {}), "", group_name);  // non-owning alias (:369-372)
        return true;
    }

    // single-pose overloads (:376-430): all forward to the multi-pose form
    bool searchPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, double timeout, std::vector<double>& solution,
                          moveit_msgs::MoveItErrorCodes& error_code, const kinematics::KinematicsQueryOptions& options = kinematics::KinematicsQueryOptions()) const override {
        return searchPositionIK(std::vector<geometry_msgs::Pose>{ik_pose}, ik_seed_state, timeout, std::vector<double>(), solution, IKCallbackFn(), error_code, options);
    }
    bool searchPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, double timeout, const std::vector<double>& consistency_limits,
                          std::vector<double>& solution, moveit_msgs::MoveItErrorCodes& error_code,
                          const kinematics::KinematicsQueryOptions& options = kinematics::KinematicsQueryOptions()) const override {
        return searchPositionIK(std::vector<geometry_msgs::Pose>{ik_pose}, ik_seed_state, timeout, consistency_limits, solution, IKCallbackFn(), error_code, options);
    }
    bool searchPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, double timeout, std::vector<double>& solution,
                          const IKCallbackFn& solution_callback, moveit_msgs::MoveItErrorCodes& error_code,
                          const kinematics::KinematicsQueryOptions& options = kinematics::KinematicsQueryOptions()) const override {
        return searchPositionIK(std::vector<geometry_msgs::Pose>{ik_pose}, ik_seed_state, timeout, std::vector<double>(), solution, solution_callback, error_code,
                                options);
    }
    bool searchPositionIK(const geometry_msgs::Pose& ik_pose, const std::vector<double>& ik_seed_state, double timeout, const std::vector<double>& consistency_limits,
                          std::vector<double>& solution, const IKCallbackFn& solution_callback, moveit_msgs::MoveItErrorCodes& error_code,
                          const kinematics::KinematicsQueryOptions& options = kinematics::KinematicsQueryOptions()) const override {
        return searchPositionIK(std::vector<geometry_msgs::Pose>{ik_pose}, ik_seed_state, timeout, consistency_limits, solution, solution_callback, error_code,
                                options);
    }

    // The batched core: n queries of one goal structure, marshalled and enqueued here, finished by finishBatch.  poses[k] (tips of query k;
    // ignored with options.replace), seeds[k] (group variables); `timeout` bounds the whole call (ik_parallel.h:160, honoured on the device).
    std::shared_ptr<bio_ik::core::Engine::Ticket> submitBatch(const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses,
                                                              const std::vector<std::vector<double>>& ik_seed_states, double timeout,
                                                              const kinematics::KinematicsQueryOptions& options, const moveit::core::RobotState* context_state) const {
        std::lock_guard<std::mutex> lock(mutex);
        if (!robot_model || !joint_model_group || !engine.ready()) throw std::runtime_error("bio_ik (MI355X): plugin not initialised");
        auto* bio_ik_options = bio_ik::toBioIKKinematicsQueryOptions(&options);
        const size_t V = robot_model->getVariableCount();
        bio_ik::core::Request rq;
        // variable default positions / context state: what the seed states are laid over (:465-472)
        rq.context.resize(V);
        if (context_state)
            for (size_t i = 0; i < V; i++) rq.context[i] = context_state->getVariablePositions()[i];
        else
            robot_model->getVariableDefaultPositions(rq.context);
        const bool replace = bio_ik_options && bio_ik_options->replace;
        if (!replace)  // all goals: defaults first, then the caller's (:550-556)
            for (auto& goal : default_goals) rq.goals.push_back(goal.get());
        if (bio_ik_options) {
            for (auto& goal : bio_ik_options->goals) rq.goals.push_back(goal.get());
            rq.fixed_joints = bio_ik_options->fixed_joints;
        }
        rq.n_pose_goals = replace ? 0 : tip_frames_.size();
        rq.seed_states = &ik_seed_states;
        // the tips are given in the base frame: its global transform takes them to the model frame (:487-502)
        Frame7 r;
        if (context_state) {
            r = toFrame(context_state->getGlobalLinkTransform(getBaseFrame()));
        } else {
            temp_state->setToDefaultValues();
            r = toFrame(temp_state->getGlobalLinkTransform(getBaseFrame()));
        }
        for (int c = 0; c < 7; c++) rq.base_frame[c] = r.v[c];
        if (rq.n_pose_goals)
            for (size_t k = 0; k < ik_seed_states.size(); k++)
                for (size_t t = 0; t < rq.n_pose_goals; t++) {
                    const Frame7 f = toFrame(ik_poses.at(k).at(t));
                    rq.tip_poses.insert(rq.tip_poses.end(), f.v, f.v + 7);
                }
        rq.timeout = timeout, rq.return_approximate_solution = options.return_approximate_solution, rq.bio = bio_ik_options;
        return engine.submit(rq);
    }
    bool finishBatch(bio_ik::core::Engine::Ticket& ticket, std::vector<std::vector<double>>& solutions, std::vector<moveit_msgs::MoveItErrorCodes>& error_codes) const {
        std::vector<uint8_t> ok;
        const bool all_ok = engine.wait(ticket, solutions, ok);  // (the wait itself needs no lock: the ticket owns its arrays)
        error_codes.assign(ok.size(), moveit_msgs::MoveItErrorCodes());
        for (size_t k = 0; k < ok.size(); k++) error_codes[k].val = ok[k] ? moveit_msgs::MoveItErrorCodes::SUCCESS : moveit_msgs::MoveItErrorCodes::NO_IK_SOLUTION;
        return all_ok;
    }
    bool solveBatch(const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses, const std::vector<std::vector<double>>& ik_seed_states, double timeout,
                    std::vector<std::vector<double>>& solutions, std::vector<moveit_msgs::MoveItErrorCodes>& error_codes,
                    const kinematics::KinematicsQueryOptions& options, const moveit::core::RobotState* context_state) const {
        auto ticket = submitBatch(ik_poses, ik_seed_states, timeout, options, context_state);
        return finishBatch(*ticket, solutions, error_codes);
    }

    // :437-655, the overload every other one forwards to
    bool searchPositionIK(const std::vector<geometry_msgs::Pose>& ik_poses, const std::vector<double>& ik_seed_state, double timeout,
                          const std::vector<double>& /*consistency_limits*/, std::vector<double>& solution, const IKCallbackFn& solution_callback,
                          moveit_msgs::MoveItErrorCodes& error_code, const kinematics::KinematicsQueryOptions& options = kinematics::KinematicsQueryOptions(),
                          const moveit::core::RobotState* context_state = nullptr) const override {
        std::vector<std::vector<double>> sols;
        std::vector<moveit_msgs::MoveItErrorCodes> codes;
        const bool ok = solveBatch({ik_poses}, {ik_seed_state}, timeout, sols, codes, options, context_state);
        if (!sols.empty() && !sols[0].empty()) solution = sols[0];
        if (!ok) {  // no accurate solution and no approximate one requested (:638-641)
            error_code.val = error_code.NO_IK_SOLUTION;
            return false;
        }
        if (solution_callback) {  // :644-649: the callback's verdict is the result
            solution_callback(ik_poses.empty() ? geometry_msgs::Pose() : ik_poses.front(), solution, error_code);
            return error_code.val == error_code.SUCCESS;
        }
        error_code.val = error_code.SUCCESS;
        return true;
    }

    bool supportsGroup(const moveit::core::JointModelGroup*, std::string* = nullptr) const override { return true; }  // :657-662
};

static const BioIKKinematicsPlugin& pluginOf(const kinematics::KinematicsBase& solver, const char* who) {
    auto* plugin = dynamic_cast<const BioIKKinematicsPlugin*>(&solver);
    if (!plugin) throw std::runtime_error(std::string(who) + ": not a bio_ik (MI355X) kinematics plugin");
    return *plugin;
}
// the additive batched entry point of the north star ("a batched searchPositionIK()"), reachable from a KinematicsBase pointer
bool searchPositionIKBatch(const kinematics::KinematicsBase& solver, const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses,
                           const std::vector<std::vector<double>>& ik_seed_states, double timeout, std::vector<std::vector<double>>& solutions,
                           std::vector<moveit_msgs::MoveItErrorCodes>& error_codes, const kinematics::KinematicsQueryOptions& options,
                           const moveit::core::RobotState* context_state) {
    return pluginOf(solver, "searchPositionIKBatch").solveBatch(ik_poses, ik_seed_states, timeout, solutions, error_codes, options, context_state);
}


struct BatchTicket::Impl {
    std::shared_ptr<bio_ik::core::Engine::Ticket> ticket;
};
BatchTicket::BatchTicket() {}
BatchTicket::~BatchTicket() {}
BatchTicket::BatchTicket(BatchTicket&&) = default;
BatchTicket& BatchTicket::operator=(BatchTicket&&) = default;
BatchTicket searchPositionIKBatchAsync(const kinematics::KinematicsBase& solver, const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses,
                                       const std::vector<std::vector<double>>& ik_seed_states, double timeout, const kinematics::KinematicsQueryOptions& options,
                                       const moveit::core::RobotState* context_state) {
    BatchTicket t;
    t.impl.reset(new BatchTicket::Impl{pluginOf(solver, "searchPositionIKBatchAsync").submitBatch(ik_poses, ik_seed_states, timeout, options, context_state)});
    return t;
}
bool searchPositionIKBatchWait(const kinematics::KinematicsBase& solver, BatchTicket& ticket, std::vector<std::vector<double>>& solutions,
                               std::vector<moveit_msgs::MoveItErrorCodes>& error_codes) {
    if (!ticket.impl || !ticket.impl->ticket) throw std::runtime_error("searchPositionIKBatchWait: empty ticket");
    const bool ok = pluginOf(solver, "searchPositionIKBatchWait").finishBatch(*ticket.impl->ticket, solutions, error_codes);
    ticket.impl.reset();
    return ok;
}

}  // namespace bio_ik_kinematics_plugin

// register plugin (:670-671)
PLUGINLIB_EXPORT_CLASS(bio_ik_kinematics_plugin::BioIKKinematicsPlugin, kinematics::KinematicsBase);

#if defined(BIOIK_STANDIN_PLUGINLIB)
// stand-in pluginlib only: the entry point a loader resolves with dlsym (class_loader's own registry does this in a ROS workspace)
extern "C" void* pluginlib_standin_create(const char* derived_type_name) {
    auto it = pluginlib::registry().find(derived_type_name);
    return it == pluginlib::registry().end() ? nullptr : it->second.second();
}
#endif
