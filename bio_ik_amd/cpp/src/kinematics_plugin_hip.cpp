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
Frame7 concat(const Frame7& a, const Frame7& b) {  // a o b (include/bio_ik/frame.h:174-187 semantics)
    const double *q = a.v + 3, *v = b.v;
    const double tx = 2 * (q[1] * v[2] - q[2] * v[1]), ty = 2 * (q[2] * v[0] - q[0] * v[2]), tz = 2 * (q[0] * v[1] - q[1] * v[0]);
    Frame7 r;
    r.v[0] = a.v[0] + v[0] + q[3] * tx + q[1] * tz - q[2] * ty;
    r.v[1] = a.v[1] + v[1] + q[3] * ty + q[2] * tx - q[0] * tz;
    r.v[2] = a.v[2] + v[2] + q[3] * tz + q[0] * ty - q[1] * tx;
    const double *p = a.v + 3, *o = b.v + 3;
    r.v[3] = p[3] * o[0] + p[0] * o[3] + p[1] * o[2] - p[2] * o[1];
    r.v[4] = p[3] * o[1] - p[0] * o[2] + p[1] * o[3] + p[2] * o[0];
    r.v[5] = p[3] * o[2] + p[0] * o[1] - p[1] * o[0] + p[2] * o[3];
    r.v[6] = p[3] * o[3] - p[0] * o[0] - p[1] * o[1] - p[2] * o[2];
    return r;
}
int solverMode(const std::string& name) {  // IKFactory names (src/ik_evolution_2.cpp:652-654); unknown -> ERROR (src/utils.h:436)
    if (name == "bio2") return BIOIK_MODE_BIO2;
    if (name == "bio2_memetic") return BIOIK_MODE_BIO2_MEMETIC;
    if (name == "bio2_memetic_l") return BIOIK_MODE_BIO2_MEMETIC_L;
    if (name == "gd_c") return BIOIK_MODE_GD_C;  // src/ik_gradient.cpp:263
    if (name == "gd") return BIOIK_MODE_GD;      // :253
    if (name == "jac") return BIOIK_MODE_JAC;    // src/ik_gradient.cpp:289
    throw std::runtime_error("bio_ik (MI355X): solver mode '" + name + "' has no device implementation");
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
    mutable std::unique_ptr<moveit::core::RobotState> temp_state;
    mutable std::vector<std::unique_ptr<bio_ik::Goal>> default_goals;
    mutable std::mutex mutex;  // the reference's instance is not re-entrant either (mutable members, :121-124); here calls serialise

    // IKParams (src/utils.h:64-85) restricted to what the device path reads, + the additive gpu_* keys
    std::string mode = "bio2_memetic";
    int random_seed = 0;
    double dpos = DBL_MAX, drot = DBL_MAX, dtwist = 1e-5;
    bool no_wipeout = false;
    int gpu_population = 128, gpu_islands = 1, gpu_max_steps = 4096, gpu_device = 0;
    std::string gpu_fk = "exact";

    std::unique_ptr<FlatModel> flat;
    std::vector<int> devices;            // kinematics.yaml `gpu_devices: "0,1,2,3"` (default: the single `gpu_device`): a batch is sharded over them
    std::vector<bioik_model*> models;    // one per device
    mutable std::map<std::string, std::vector<bioik_problem*>> problems;  // per goal structure: one compiled problem per device

    BioIKKinematicsPlugin() {}
    ~BioIKKinematicsPlugin() override { release(); }
    void release() {
        for (auto& kv : problems)
            for (auto* p : kv.second) bioik_problem_destroy(p);
        problems.clear();
        for (auto* m : models) bioik_model_destroy(m);
        models.clear();
    }

    const std::vector<std::string>& getJointNames() const override { return joint_names; }  // :130-133
    const std::vector<std::string>& getLinkNames() const override { return link_names; }    // :135-138
    bool getPositionFK(const std::vector<std::string>&, const std::vector<double>&, std::vector<geometry_msgs::Pose>&) const override { return false; }  // :140-145
    bool getPositionIK(const geometry_msgs::Pose&, const std::vector<double>&, std::vector<double>&, moveit_msgs::MoveItErrorCodes&,
                       const kinematics::KinematicsQueryOptions& = kinematics::KinematicsQueryOptions()) const override {
        return false;  // :147-155
    }

    // :191-335
    bool load(const moveit::core::RobotModelConstPtr& model_ptr, const std::string& /*robot_description*/, const std::string& group_name) {
        if (!model_ptr) {
            // the reference parses URDF + SRDF from the parameter server here (rdf_loader, :167-189); the MoveIt versions this plugin
            // targets hand over the RobotModel (initialize(const RobotModel&, ...)), and that is the overload this build supports
            throw std::runtime_error("bio_ik (MI355X): initialize needs the moveit::core::RobotModel overload");
        }
        robot_model = model_ptr;
        joint_model_group = robot_model->getJointModelGroup(group_name);
        if (!joint_model_group) return false;  // "failed to get joint model group" (:215-218)
        joint_names.clear();
        for (auto* joint_model : joint_model_group->getJointModels())
            if (joint_model->getName() != base_frame_ && joint_model->getType() != moveit::core::JointModel::UNKNOWN &&
                joint_model->getType() != moveit::core::JointModel::FIXED)
                joint_names.push_back(joint_model->getName());
        auto tips2 = tip_frames_;
        joint_model_group->getEndEffectorTips(tips2);
        if (!tips2.empty()) tip_frames_ = tips2;
        link_names = tip_frames_;

        lookupParam("mode", mode, std::string("bio2_memetic"));
        lookupParam("random_seed", random_seed, static_cast<int>(std::random_device()()));
        lookupParam("dpos", dpos, DBL_MAX);
        lookupParam("drot", drot, DBL_MAX);
        lookupParam("dtwist", dtwist, 1e-5);
        lookupParam("no_wipeout", no_wipeout, false);
        lookupParam("gpu_population", gpu_population, 128);   // children per species and generation (reference: 16, ik_evolution_2.cpp:138)
        lookupParam("gpu_islands", gpu_islands, 1);
        lookupParam("gpu_max_steps", gpu_max_steps, 4096);    // safety cap; the caller's timeout is what normally ends a query
        lookupParam("gpu_fk", gpu_fk, std::string("exact"));  // "exact" | "linear" (the reference's linearised phenotypes)
        lookupParam("gpu_device", gpu_device, 0);
        std::string gpu_devices;
        lookupParam("gpu_devices", gpu_devices, std::string());
        devices.clear();
        {
            std::stringstream ss(gpu_devices);
            for (std::string item; std::getline(ss, item, ',');)
                if (!item.empty()) devices.push_back(std::stoi(item));
        }
        if (devices.empty()) devices.push_back(gpu_device);
        solverMode(mode);

        temp_state.reset(new moveit::core::RobotState(robot_model));
        flat.reset(new FlatModel(*robot_model));
        bioik_model_desc md = flat->desc();
        release();
        for (int dev : devices) {
            bioik_model* m = nullptr;
            if (bioik_model_create(&md, dev, &m) != BIOIK_OK) throw std::runtime_error(std::string("bio_ik (MI355X): ") + bioik_last_error());
            models.push_back(m);
        }

        default_goals.clear();  // :279-329
        for (size_t i = 0; i < tip_frames_.size(); i++) {
            auto* goal = new bio_ik::PoseGoal();
            goal->setLinkName(tip_frames_[i]);
            double rotation_scale = 0.5;
            lookupParam("rotation_scale", rotation_scale, rotation_scale);
            bool position_only_ik = false;
            lookupParam("position_only_ik", position_only_ik, position_only_ik);
            if (position_only_ik) rotation_scale = 0;
            goal->setRotationScale(rotation_scale);
            default_goals.emplace_back(goal);
        }
        double weight = 0;
        lookupParam("center_joints_weight", weight, 0.0);
        if (weight > 0.0) default_goals.emplace_back(new bio_ik::CenterJointsGoal(weight));
        weight = 0;
        lookupParam("avoid_joint_limits_weight", weight, 0.0);
        if (weight > 0.0) default_goals.emplace_back(new bio_ik::AvoidJointLimitsGoal(weight));
        weight = 0;
        lookupParam("minimal_displacement_weight", weight, 0.0);
        if (weight > 0.0) default_goals.emplace_back(new bio_ik::MinimalDisplacementGoal(weight));
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

    // one compiled problem (Problem::initialize + RobotFK::initialize on the device side) per goal STRUCTURE; the numbers travel per query
    const std::vector<bioik_problem*>& problemFor(const std::vector<const bio_ik::Goal*>& goals, const std::vector<std::string>& fixed) const {
        std::ostringstream key;
        key << std::hexfloat;
        for (auto* g : goals) key << g->gpuOpcode() << ':' << g->gpuLinkName() << ':' << g->gpuVariableName() << ':' << g->getWeight() << ':' << g->isSecondary() << ';';
        for (auto& f : fixed) key << '#' << f;
        auto it = problems.find(key.str());
        if (it != problems.end()) return it->second;
        std::vector<bioik_goal_desc> gd;
        for (auto* g : goals) {
            if (g->gpuOpcode() < 0) throw std::runtime_error("bio_ik (MI355X): goal type without a device implementation (host-callback goal)");
            bioik_goal_desc d{g->gpuOpcode(), -1, -1, g->isSecondary() ? 1 : 0, g->getWeight()};
            if (!g->gpuLinkName().empty()) {
                auto* l = robot_model->getLinkModel(g->gpuLinkName());
                if (!l) throw std::runtime_error("link not found: " + g->gpuLinkName());  // problem.cpp:141
                d.link = (int32_t)l->getLinkIndex();
            }
            if (!g->gpuVariableName().empty()) d.variable = robot_model->getVariableIndex(g->gpuVariableName());
            gd.push_back(d);
        }
        std::vector<int32_t> group_joints, fixed_idx;
        for (auto* j : joint_model_group->getActiveJointModels()) group_joints.push_back((int32_t)j->getChildLinkModel()->getLinkIndex());
        for (auto& f : fixed) {
            auto* j = robot_model->getJointModel(f);
            if (!j) throw std::runtime_error("joint not found: " + f);
            fixed_idx.push_back((int32_t)j->getChildLinkModel()->getLinkIndex());
        }
        bioik_problem_desc pd{};
        pd.struct_size = sizeof(pd);
        pd.n_group_joints = (uint32_t)group_joints.size(), pd.group_joints = group_joints.data();
        pd.n_goals = (uint32_t)gd.size(), pd.goals = gd.data();
        pd.n_fixed_joints = (uint32_t)fixed_idx.size(), pd.fixed_joints = fixed_idx.data();
        std::vector<bioik_problem*> per_device;
        for (auto* m : models) {
            bioik_problem* p = nullptr;
            if (bioik_problem_create(m, &pd, &p) != BIOIK_OK) {
                for (auto* q : per_device) bioik_problem_destroy(q);
                throw std::runtime_error(std::string("bio_ik (MI355X): ") + bioik_last_error());
            }
            per_device.push_back(p);
        }
        return problems[key.str()] = per_device;
    }

    // The batched core: n queries of one goal structure, ONE bioik_solve_batch call.  poses[k] (tips of query k; ignored with
    // options.replace), seeds[k] (group variables); `timeout` bounds the whole call (ik_parallel.h:160, honoured on the device).
    bool solveBatch(const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses, const std::vector<std::vector<double>>& ik_seed_states, double timeout,
                    std::vector<std::vector<double>>& solutions, std::vector<moveit_msgs::MoveItErrorCodes>& error_codes,
                    const kinematics::KinematicsQueryOptions& options, const moveit::core::RobotState* context_state) const {
        std::lock_guard<std::mutex> lock(mutex);
        if (!robot_model || models.empty()) throw std::runtime_error("bio_ik (MI355X): plugin not initialised");
        auto* bio_ik_options = bio_ik::toBioIKKinematicsQueryOptions(&options);
        const size_t n = ik_seed_states.size(), V = robot_model->getVariableCount();
        // get variable default positions / context state, overwrite used variables with seed state (:465-485)
        state.resize(V);
        if (context_state)
            for (size_t i = 0; i < V; i++) state[i] = context_state->getVariablePositions()[i];
        else
            robot_model->getVariableDefaultPositions(state);
        std::vector<double> seeds(n * V);
        for (size_t k = 0; k < n; k++) {
            for (size_t v = 0; v < V; v++) seeds[k * V + v] = state[v];
            size_t i = 0;
            for (auto& joint_name : getJointNames()) {
                auto* joint_model = robot_model->getJointModel(joint_name);
                if (!joint_model) continue;
                for (size_t vi = 0; vi < joint_model->getVariableCount(); vi++) seeds[k * V + joint_model->getFirstVariableIndex() + vi] = ik_seed_states[k].at(i++);
            }
        }
        const bool replace = bio_ik_options && bio_ik_options->replace;
        // all goals: defaults first, then the caller's (:550-556)
        std::vector<const bio_ik::Goal*> all_goals;
        if (!replace)
            for (auto& goal : default_goals) all_goals.push_back(goal.get());
        if (bio_ik_options)
            for (auto& goal : bio_ik_options->goals) all_goals.push_back(goal.get());
        const std::vector<bioik_problem*>& shards = problemFor(all_goals, bio_ik_options ? bio_ik_options->fixed_joints : std::vector<std::string>());
        bioik_problem* problem = shards.front();
        const size_t P = (size_t)bioik_problem_param_count(problem);
        // transform tips to the model frame (:487-502) and let every goal write its numbers
        Frame7 r;
        if (context_state) {
            r = toFrame(context_state->getGlobalLinkTransform(getBaseFrame()));
        } else {
            temp_state->setToDefaultValues();
            r = toFrame(temp_state->getGlobalLinkTransform(getBaseFrame()));
        }
        std::vector<double> params(n * P), row;
        for (size_t k = 0; k < n; k++) {
            row.clear();
            size_t gi = 0;
            for (auto* g : all_goals) {
                if (!replace && gi < tip_frames_.size()) {
                    const Frame7 m = concat(r, toFrame(ik_poses.at(k).at(gi)));
                    auto* goal = static_cast<bio_ik::PoseGoal*>(const_cast<bio_ik::Goal*>(g));
                    goal->setPosition(bio_ik::Vector3(m.v[0], m.v[1], m.v[2]));
                    goal->setOrientation(bio_ik::Quaternion(m.v[3], m.v[4], m.v[5], m.v[6]));  // normalises (goal_types.h:146)
                }
                g->gpuParams(row);
                gi++;
            }
            for (size_t i = 0; i < P; i++) params[k * P + i] = row.at(i);
        }
        bioik_solve_params sp;
        bioik_default_solve_params(&sp);
        sp.mode = solverMode(mode);
        sp.fk_mode = gpu_fk == "linear" ? BIOIK_FK_LINEAR : BIOIK_FK_EXACT;
        sp.population = gpu_population, sp.islands = gpu_islands, sp.max_steps = gpu_max_steps;
        sp.random_seed = (uint64_t)(uint32_t)random_seed;
        sp.dpos = dpos, sp.drot = drot, sp.dtwist = dtwist;
        sp.no_wipeout = no_wipeout ? 1 : 0;
        sp.timeout = timeout > 0.0 ? timeout : 0.0;  // problem.timeout = t0 + timeout (:504); the budget starts with the launch
        std::vector<double> sol(n * V), fit(n);
        std::vector<int32_t> suc(n), steps(n);
        solutions.assign(n, std::vector<double>());
        error_codes.assign(n, moveit_msgs::MoveItErrorCodes());
        // one call for the whole batch: a single device, or contiguous shards over the configured devices (no exchange between shards)
        const int rc = shards.size() == 1 ? bioik_solve_batch(problem, &sp, n, seeds.data(), params.data(), sol.data(), fit.data(), suc.data(), steps.data())
                                          : bioik_solve_batch_multi(shards.data(), (int)shards.size(), &sp, n, seeds.data(), params.data(), sol.data(), fit.data(),
                                                                    suc.data(), steps.data());
        if (rc != BIOIK_OK) {
            for (auto& e : error_codes) e.val = moveit_msgs::MoveItErrorCodes::NO_IK_SOLUTION;  // device errors never abort the caller
            return false;
        }
        std::vector<int32_t> active((size_t)bioik_problem_active_variable_count(problem));
        bioik_problem_active_variables(problem, active.data());
        bool all_ok = true;
        for (size_t k = 0; k < n; k++) {
            double* st = &sol[k * V];
            // wrap angles (:580-613)
            for (int ivar : active) {
                double v = st[ivar];
                const moveit::core::JointModel* jm = robot_model->getJointOfVariable(ivar);
                if (jm->getType() == moveit::core::JointModel::REVOLUTE && robot_model->getMimicJointModels().empty()) {
                    const moveit::core::VariableBounds& b = jm->getVariableBounds()[(size_t)(ivar - jm->getFirstVariableIndex())];
                    const double rr = seeds[k * V + ivar], lo = b.min_position_, hi = b.max_position_;
                    if (rr < v - M_PI || rr > v + M_PI) {  // move close to initial guess
                        v -= rr, v /= (2 * M_PI), v += 0.5, v -= std::floor(v), v -= 0.5, v *= (2 * M_PI), v += rr;
                    }
                    if (v > hi) v -= std::ceil(std::max(0.0, v - hi) / (2 * M_PI)) * (2 * M_PI);  // wrap at joint limits
                    if (v < lo) v += std::ceil(std::max(0.0, lo - v) / (2 * M_PI)) * (2 * M_PI);
                    if (v < lo) v = lo;  // clamp at edges
                    if (v > hi) v = hi;
                }
                st[ivar] = v;
            }
            robot_model->enforcePositionBounds(st);  // :616
            for (auto& joint_name : getJointNames()) {  // map result to jointgroup variables (:619-629)
                auto* joint_model = robot_model->getJointModel(joint_name);
                if (!joint_model) continue;
                for (size_t vi = 0; vi < joint_model->getVariableCount(); vi++) solutions[k].push_back(st[joint_model->getFirstVariableIndex() + vi]);
            }
            const bool ok = suc[k] || options.return_approximate_solution;  // :638-641
            error_codes[k].val = ok ? moveit_msgs::MoveItErrorCodes::SUCCESS : moveit_msgs::MoveItErrorCodes::NO_IK_SOLUTION;
            all_ok = all_ok && ok;
        }
        if (bio_ik_options && n) bio_ik_options->solution_fitness = fit[n - 1];  // :632-634
        return all_ok;
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

// the additive batched entry point of the north star ("a batched searchPositionIK()"), reachable from a KinematicsBase pointer
bool searchPositionIKBatch(const kinematics::KinematicsBase& solver, const std::vector<std::vector<geometry_msgs::Pose>>& ik_poses,
                           const std::vector<std::vector<double>>& ik_seed_states, double timeout, std::vector<std::vector<double>>& solutions,
                           std::vector<moveit_msgs::MoveItErrorCodes>& error_codes, const kinematics::KinematicsQueryOptions& options,
                           const moveit::core::RobotState* context_state) {
    auto* plugin = dynamic_cast<const BioIKKinematicsPlugin*>(&solver);
    if (!plugin) throw std::runtime_error("searchPositionIKBatch: not a bio_ik (MI355X) kinematics plugin");
    return plugin->solveBatch(ik_poses, ik_seed_states, timeout, solutions, error_codes, options, context_state);
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
