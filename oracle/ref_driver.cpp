// ref_driver.cpp — ORACLE SUPPORT (test infrastructure): a C API over the REFERENCE'S OWN classes.
//
// Compiled together with /root/reference/src/problem.cpp and /root/reference/src/ik_evolution_2.cpp (unmodified, from
// where they lie) against the stand-in third-party headers of oracle/ref_shim into oracle/_ref/libbioik_ref.so
// (`make -C oracle ref`).  Nothing here restates bio_ik: every number returned is computed by the reference's code —
// bio_ik::Frame algebra (include/bio_ik/frame.h), RobotFK (src/forward_kinematics.h), Problem / Goal::evaluate
// (src/problem.cpp, include/bio_ik/goal_types.h) and IKEvolution2 (src/ik_evolution_2.cpp) through IKFactory.
// The driver only marshals the flat model / problem PODs of include/bioik_hip.h into the MoveIt-shaped containers the
// reference reads, and goal opcodes + numbers into the reference's goal objects.
#include "ref_prelude.h"

#include "ik_base.h"

#include <chrono>
#include <bio_ik/goal_types.h>

#include "../include/bioik_hip.h"

using namespace bio_ik;

// The options struct's constructor / destructor live in src/kinematics_plugin.cpp:73-101 (a MoveIt translation unit that is
// not part of this build), where they also maintain the plugin's pointer registry; without the plugin only the member
// initialisation matters.
namespace bio_ik {
BioIKKinematicsQueryOptions::BioIKKinematicsQueryOptions() : replace(false), solution_fitness(0) {}
BioIKKinematicsQueryOptions::~BioIKKinematicsQueryOptions() {}
}  // namespace bio_ik

namespace {
thread_local std::string g_err;

std::shared_ptr<moveit::core::RobotModel> build_model(const bioik_model_desc& d) {
    auto m = std::make_shared<moveit::core::RobotModel>();
    std::vector<moveit::core::JointModel*> joints(d.n_links);
    std::vector<moveit::core::LinkModel*> links(d.n_links);
    // variable names: "v<i>"
    for (uint32_t v = 0; v < d.n_variables; v++) {
        m->variable_names_.push_back("v" + std::to_string(v));
        m->variable_index_[m->variable_names_.back()] = (int)v;
    }
    m->joint_of_variable_.assign(d.n_variables, nullptr);
    for (uint32_t i = 0; i < d.n_links; i++) {
        moveit::core::JointModel* j = nullptr;
        int nv = 0;
        switch (d.joint_type[i]) {
            case BIOIK_JOINT_REVOLUTE: {
                auto* r = new moveit::core::RevoluteJointModel();
                r->axis_ = Eigen::Vector3d(d.joint_axis[3 * i], d.joint_axis[3 * i + 1], d.joint_axis[3 * i + 2]);
                r->type_ = moveit::core::JointModel::REVOLUTE;
                j = r, nv = 1;
                break;
            }
            case BIOIK_JOINT_PRISMATIC: {
                auto* r = new moveit::core::PrismaticJointModel();
                r->axis_ = Eigen::Vector3d(d.joint_axis[3 * i], d.joint_axis[3 * i + 1], d.joint_axis[3 * i + 2]);
                r->type_ = moveit::core::JointModel::PRISMATIC;
                j = r, nv = 1;
                break;
            }
            case BIOIK_JOINT_FLOATING: j = new moveit::core::FloatingJointModel(), j->type_ = moveit::core::JointModel::FLOATING, nv = 7; break;
            case BIOIK_JOINT_PLANAR: j = new moveit::core::PlanarJointModel(), j->type_ = moveit::core::JointModel::PLANAR, nv = 3; break;
            default: j = new moveit::core::FixedJointModel(), j->type_ = moveit::core::JointModel::FIXED; break;
        }
        j->name_ = "j" + std::to_string(i);
        j->joint_index_ = (int)i;
        j->first_variable_index_ = nv ? d.joint_first_variable[i] : 0;
        for (int k = 0; k < nv; k++) {
            int v = d.joint_first_variable[i] + k;
            j->variable_names_.push_back(m->variable_names_[v]);
            moveit::core::VariableBounds b;
            b.min_position_ = d.var_min[v], b.max_position_ = d.var_max[v], b.position_bounded_ = d.var_bounded[v] != 0;
            b.max_velocity_ = d.var_max_velocity[v], b.min_velocity_ = -d.var_max_velocity[v], b.velocity_bounded_ = d.var_max_velocity[v] > 0;
            j->variable_bounds_.push_back(b);
            m->joint_of_variable_[v] = j;
        }
        auto* l = new moveit::core::LinkModel();
        l->name_ = "l" + std::to_string(i);
        l->link_index_ = (int)i;
        l->parent_joint_ = j;
        l->parent_link_ = d.link_parent[i] >= 0 ? links[d.link_parent[i]] : nullptr;
        const double* o = d.link_origin + 7 * i;
        l->joint_origin_transform_.t = Eigen::Vector3d(o[0], o[1], o[2]);
        l->joint_origin_transform_.r = Eigen::Quaterniond(o[6], o[3], o[4], o[5]).toRotationMatrix();
        j->child_link_ = l;
        j->parent_link_ = l->parent_link_;
        joints[i] = j, links[i] = l;
        m->joints_.emplace_back(j), m->links_.emplace_back(l);
        m->joint_ptrs_.push_back(j), m->link_ptrs_.push_back(l);
        m->joint_names_.push_back(j->name_), m->link_names_.push_back(l->name_);
        auto ul = std::make_shared<urdf::Link>();  // the URDF side of the link: its <inertial>, if it has one
        if (d.link_mass && d.link_mass[i] != 0.0) {
            ul->inertial = std::make_shared<urdf::Inertial>();
            ul->inertial->mass = d.link_mass[i];
            ul->inertial->origin.position.x = d.link_center[3 * i], ul->inertial->origin.position.y = d.link_center[3 * i + 1];
            ul->inertial->origin.position.z = d.link_center[3 * i + 2];
        }
        m->urdf_->links_[l->name_] = ul;
    }
    for (uint32_t i = 0; i < d.n_links; i++) {
        if (d.joint_mimic && d.joint_mimic[i] >= 0) {
            joints[i]->mimic_ = joints[d.joint_mimic[i]];
            joints[i]->mimic_factor_ = d.joint_mimic_factor[i], joints[i]->mimic_offset_ = d.joint_mimic_offset[i];
            m->mimic_joints_.push_back(joints[i]);
        }
        if (joints[i]->type_ != moveit::core::JointModel::FIXED && !joints[i]->mimic_) m->active_joints_.push_back(joints[i]);
    }
    return m;
}

struct Ref {
    std::shared_ptr<moveit::core::RobotModel> model;
    moveit::core::JointModelGroup* group = nullptr;
    std::vector<bioik_goal_desc> goal_descs;
    std::vector<int> param_off;
    int P = 0;
    BioIKKinematicsQueryOptions options;
    bool have_options = false;
    IKParams ikparams;
    std::vector<std::unique_ptr<Goal>> goals;  // rebuilt per query
    Problem problem;
    std::unique_ptr<IKSolver> batch_solver;  // ref_solve_batch: one solver object for all queries

    static tf2::Vector3 v3(const double* p) { return tf2::Vector3(p[0], p[1], p[2]); }
    // the reference's goal object for one opcode (constructors as in include/bio_ik/goal_types.h)
    Goal* make_goal(const bioik_goal_desc& g, const double* p) {
        std::string link = g.link >= 0 ? model->link_names_[g.link] : std::string();
        bool sec = g.secondary != 0;
        auto need_primary = [&]() {
            if (sec) throw std::runtime_error("the reference's link goals cannot be marked secondary");
        };
        switch (g.type) {
            case BIOIK_GOAL_POSITION: need_primary(); return new PositionGoal(link, v3(p), g.weight);
            case BIOIK_GOAL_ORIENTATION: need_primary(); return new OrientationGoal(link, tf2::Quaternion(p[0], p[1], p[2], p[3]), g.weight);
            case BIOIK_GOAL_POSE: {
                need_primary();
                auto* x = new PoseGoal(link, v3(p), tf2::Quaternion(p[3], p[4], p[5], p[6]), g.weight);
                x->setRotationScale(p[7]);
                return x;
            }
            case BIOIK_GOAL_LOOK_AT: need_primary(); return new LookAtGoal(link, v3(p), v3(p + 3), g.weight);
            case BIOIK_GOAL_MAX_DISTANCE: need_primary(); return new MaxDistanceGoal(link, v3(p), p[3], g.weight);
            case BIOIK_GOAL_MIN_DISTANCE: need_primary(); return new MinDistanceGoal(link, v3(p), p[3], g.weight);
            case BIOIK_GOAL_LINE: need_primary(); return new LineGoal(link, v3(p), v3(p + 3), g.weight);
            case BIOIK_GOAL_PLANE: need_primary(); return new PlaneGoal(link, v3(p), v3(p + 3), g.weight);
            case BIOIK_GOAL_AVOID_JOINT_LIMITS: return new AvoidJointLimitsGoal(g.weight, sec);
            case BIOIK_GOAL_CENTER_JOINTS: return new CenterJointsGoal(g.weight, sec);
            case BIOIK_GOAL_REGULARIZATION:
                if (sec) throw std::runtime_error("RegularizationGoal has no secondary flag in the reference");
                return new RegularizationGoal(g.weight);
            case BIOIK_GOAL_MINIMAL_DISPLACEMENT: return new MinimalDisplacementGoal(g.weight, sec);
            case BIOIK_GOAL_JOINT_VARIABLE: return new JointVariableGoal(model->variable_names_[g.variable], p[0], g.weight, sec);
            case BIOIK_GOAL_SIDE: need_primary(); return new SideGoal(link, v3(p), v3(p + 3), g.weight);
            case BIOIK_GOAL_DIRECTION: need_primary(); return new DirectionGoal(link, v3(p), v3(p + 3), g.weight);
            case BIOIK_GOAL_CONE: need_primary(); return new ConeGoal(link, v3(p), p[3], v3(p + 4), v3(p + 7), p[10], g.weight);
            case BIOIK_GOAL_BALANCE: {
                need_primary();
                auto* x = new BalanceGoal(v3(p), g.weight);
                x->setAxis(v3(p + 3));
                return x;
            }
        }
        throw std::runtime_error("unknown goal opcode");
    }
    void set_query(const double* seed, const double* params) {
        goals.clear();
        std::vector<const Goal*> gp;
        for (size_t i = 0; i < goal_descs.size(); i++) {
            goals.emplace_back(make_goal(goal_descs[i], params + param_off[i]));
            gp.push_back(goals.back().get());
        }
        problem.initial_guess.assign(seed, seed + model->getVariableCount());
        problem.initialize(model, group, ikparams, gp, have_options ? &options : nullptr);  // src/problem.cpp:72-228
    }
};

int goal_np(int t) { return bioik_goal_param_count(t); }
void frame_out(const Frame& f, double* o) {
    o[0] = f.pos.x(), o[1] = f.pos.y(), o[2] = f.pos.z(), o[3] = f.rot.x(), o[4] = f.rot.y(), o[5] = f.rot.z(), o[6] = f.rot.w();
}
Frame frame_in(const double* p) { return Frame(tf2::Vector3(p[0], p[1], p[2]), tf2::Quaternion(p[3], p[4], p[5], p[6])); }
std::vector<double> full_vars(const Ref& r, const double* seed, const double* genes) {
    std::vector<double> v(seed, seed + r.model->getVariableCount());
    for (size_t i = 0; i < r.problem.active_variables.size(); i++) v[r.problem.active_variables[i]] = genes[i];
    return v;
}
}  // namespace

// bioik_goal_param_count lives in the product library; the reference driver carries its own copy of the table
extern "C" int bioik_goal_param_count(int t) {
    static const int n[17] = {3, 4, 8, 6, 4, 4, 6, 6, 0, 0, 0, 0, 1, 6, 6, 11, 6};
    return (t >= 0 && t < 17) ? n[t] : -1;
}

#define TRY try {
#define CATCH(ret)                 \
    }                              \
    catch (const std::exception& e) { \
        g_err = e.what();          \
        return ret;                \
    }

extern "C" {
const char* ref_last_error(void) { return g_err.c_str(); }

void* ref_create(const bioik_model_desc* md, const bioik_problem_desc* pd, const bioik_solve_params* sp) {
    TRY
    auto* r = new Ref();
    r->model = build_model(*md);
    auto* g = new moveit::core::JointModelGroup();
    g->name_ = "group";
    g->parent_model_ = r->model.get();
    for (uint32_t i = 0; i < pd->n_group_joints; i++) {
        const auto* j = r->model->joint_ptrs_[pd->group_joints[i]];
        g->active_joint_models_.push_back(j), g->joint_models_.push_back(j), g->joint_model_names_.push_back(j->getName());
        for (auto& n : j->getVariableNames()) g->variable_names_.push_back(n);
    }
    r->model->groups_["group"].reset(g);
    r->group = g;
    for (uint32_t i = 0; i < pd->n_goals; i++) {
        r->goal_descs.push_back(pd->goals[i]);
        r->param_off.push_back(r->P);
        r->P += goal_np(pd->goals[i].type);
    }
    if (pd->n_fixed_joints) {
        r->have_options = true;
        for (uint32_t i = 0; i < pd->n_fixed_joints; i++) r->options.fixed_joints.push_back(r->model->joint_names_[pd->fixed_joints[i]]);
    }
    r->ikparams.robot_model = r->model;
    r->ikparams.joint_model_group = g;
    r->ikparams.solver_class_name = sp && sp->mode == BIOIK_MODE_BIO2 ? "bio2" : (sp && sp->mode == BIOIK_MODE_BIO2_MEMETIC_L ? "bio2_memetic_l" : "bio2_memetic");
    if (sp && sp->mode == BIOIK_MODE_GD_C) r->ikparams.solver_class_name = "gd_c";  // src/ik_gradient.cpp:263
    if (sp && sp->mode == BIOIK_MODE_GD) r->ikparams.solver_class_name = "gd";      // :253
    if (sp && sp->mode == BIOIK_MODE_GD_R) r->ikparams.solver_class_name = "gd_r";  // :258
    if (sp && sp->mode == BIOIK_MODE_JAC) r->ikparams.solver_class_name = "jac";    // src/ik_gradient.cpp:289
    r->ikparams.enable_counter = false;
    r->ikparams.thread_count = 1;
    r->ikparams.random_seed = sp ? (int)sp->random_seed : 0;
    auto thr = [](double v) { return (v < 0.0 || !(v < FLT_MAX)) ? DBL_MAX : v; };
    r->ikparams.dpos = sp ? thr(sp->dpos) : DBL_MAX, r->ikparams.drot = sp ? thr(sp->drot) : DBL_MAX, r->ikparams.dtwist = sp ? thr(sp->dtwist) : 1e-5;
    r->ikparams.opt_no_wipeout = sp ? sp->no_wipeout != 0 : false;
    r->ikparams.population_size = 8, r->ikparams.elite_count = 4, r->ikparams.linear_fitness = false;
    // structure of the problem (tips, active variables) from a neutral query
    std::vector<double> seed(md->n_variables, 0.0), params((size_t)r->P + 1, 0.0);
    for (size_t i = 0; i < r->goal_descs.size(); i++) {  // unit quaternions / axes so that constructors do not divide by zero
        double* p = params.data() + r->param_off[i];
        switch (r->goal_descs[i].type) {
            case BIOIK_GOAL_ORIENTATION: p[3] = 1; break;
            case BIOIK_GOAL_POSE: p[6] = 1, p[7] = 0.5; break;
            case BIOIK_GOAL_LINE: case BIOIK_GOAL_PLANE: p[5] = 1; break;
            default: break;
        }
    }
    r->set_query(seed.data(), params.data());
    return r;
    CATCH(nullptr)
}
void ref_destroy(void* h) { delete (Ref*)h; }

int ref_info(void* h, int32_t* out4) {
    Ref& r = *(Ref*)h;
    out4[0] = (int32_t)r.problem.active_variables.size(), out4[1] = (int32_t)r.problem.tip_link_indices.size(), out4[2] = r.P,
    out4[3] = (int32_t)r.model->getVariableCount();
    return 0;
}
int ref_active_variables(void* h, int32_t* out) {
    Ref& r = *(Ref*)h;
    for (size_t i = 0; i < r.problem.active_variables.size(); i++) out[i] = (int32_t)r.problem.active_variables[i];
    return 0;
}
int ref_tip_links(void* h, int32_t* out) {
    Ref& r = *(Ref*)h;
    for (size_t i = 0; i < r.problem.tip_link_indices.size(); i++) out[i] = (int32_t)r.problem.tip_link_indices[i];
    return 0;
}
// RobotInfo (include/bio_ik/robot_info.h) per variable: clip_min clip_max span min max max_velocity_rcp
int ref_robot_info(void* h, double* out) {
    Ref& r = *(Ref*)h;
    RobotInfo info(r.model);
    for (size_t v = 0; v < r.model->getVariableCount(); v++) {
        double* o = out + v * 6;
        o[0] = info.getClipMin(v), o[1] = info.getClipMax(v), o[2] = info.getSpan(v), o[3] = info.getMin(v), o[4] = info.getMax(v), o[5] = info.getMaxVelocityRcp(v);
    }
    return 0;
}
// the goal numbers as the reference's goal objects store them (constructors / setters normalise some of them)
int ref_canonical_params(void* h, const double* params, double* out) {
    TRY
    Ref& r = *(Ref*)h;
    for (size_t i = 0; i < r.goal_descs.size(); i++) {
        const double* p = params + r.param_off[i];
        double* o = out + r.param_off[i];
        std::unique_ptr<Goal> g(r.make_goal(r.goal_descs[i], p));
        for (int k = 0; k < goal_np(r.goal_descs[i].type); k++) o[k] = p[k];
        if (auto* x = dynamic_cast<PoseGoal*>(g.get())) o[3] = x->getOrientation().x(), o[4] = x->getOrientation().y(), o[5] = x->getOrientation().z(), o[6] = x->getOrientation().w();
        if (auto* x = dynamic_cast<OrientationGoal*>(g.get())) o[0] = x->getOrientation().x(), o[1] = x->getOrientation().y(), o[2] = x->getOrientation().z(), o[3] = x->getOrientation().w();
        if (auto* x = dynamic_cast<LineGoal*>(g.get())) o[3] = x->getDirection().x(), o[4] = x->getDirection().y(), o[5] = x->getDirection().z();
        if (auto* x = dynamic_cast<PlaneGoal*>(g.get())) o[3] = x->getNormal().x(), o[4] = x->getNormal().y(), o[5] = x->getNormal().z();
    }
    return 0;
    CATCH(-1)
}

// ---- L1: include/bio_ik/frame.h ----
void ref_quat_mul_vec(const double* q, const double* v, double* out) {
    tf2::Vector3 r;
    quat_mul_vec(tf2::Quaternion(q[0], q[1], q[2], q[3]), tf2::Vector3(v[0], v[1], v[2]), r);
    out[0] = r.x(), out[1] = r.y(), out[2] = r.z();
}
void ref_quat_mul_quat(const double* p, const double* q, double* out) {
    tf2::Quaternion r;
    quat_mul_quat(tf2::Quaternion(p[0], p[1], p[2], p[3]), tf2::Quaternion(q[0], q[1], q[2], q[3]), r);
    out[0] = r.x(), out[1] = r.y(), out[2] = r.z(), out[3] = r.w();
}
void ref_frame_concat(const double* a, const double* b, double* out) {
    Frame r;
    concat(frame_in(a), frame_in(b), r);
    frame_out(r, out);
}
void ref_frame_invert(const double* a, double* out) {
    Frame r;
    invert(frame_in(a), r);
    frame_out(r, out);
}
void ref_frame_change(const double* a, const double* b, const double* c, double* out) {
    Frame r;
    change(frame_in(a), frame_in(b), frame_in(c), r);
    frame_out(r, out);
}
void ref_normalize_fast(double* q) {
    tf2::Quaternion x(q[0], q[1], q[2], q[3]);
    normalizeFast(x);
    q[0] = x.x(), q[1] = x.y(), q[2] = x.z(), q[3] = x.w();
}
void ref_frame_twist(const double* a, const double* b, double* out6) {
    KDL::Twist t = frameTwist(frame_in(a), frame_in(b));
    for (int i = 0; i < 6; i++) out6[i] = t(i);
}

// ---- L2: src/forward_kinematics.h ----
int ref_fk(void* h, size_t n, const double* vars, double* tips) {
    TRY
    Ref& r = *(Ref*)h;
    size_t V = r.model->getVariableCount(), T = r.problem.tip_link_indices.size();
    RobotFK fk(r.model);
    fk.initialize(r.problem.tip_link_indices);
    for (size_t k = 0; k < n; k++) {
        std::vector<double> v(vars + k * V, vars + (k + 1) * V);
        fk.applyConfiguration(v);
        for (size_t t = 0; t < T; t++) frame_out(fk.getTipFrames()[t], tips + (k * T + t) * 7);
    }
    return 0;
    CATCH(-1)
}
// mutation approximator around base_genes: tip frames [T][7], linear phenotypes of n genotypes [n][T][7]
int ref_approx_eval(void* h, const double* seed, const double* base_genes, size_t n, const double* genes, double* base_tips, double* frames) {
    TRY
    Ref& r = *(Ref*)h;
    size_t D = r.problem.active_variables.size(), T = r.problem.tip_link_indices.size();
    RobotFK fk(r.model);
    fk.initialize(r.problem.tip_link_indices);
    fk.applyConfiguration(full_vars(r, seed, base_genes));
    fk.initializeMutationApproximator(r.problem.active_variables);
    for (size_t t = 0; t < T; t++) frame_out(fk.getTipFrames()[t], base_tips + t * 7);
    std::vector<const double*> ptrs(n);
    for (size_t k = 0; k < n; k++) ptrs[k] = genes + k * D;
    std::vector<aligned_vector<Frame>> out;
    fk.computeApproximateMutations(n, ptrs.data(), out);
    for (size_t k = 0; k < n; k++)
        for (size_t t = 0; t < T; t++) frame_out(out[k][t], frames + (k * T + t) * 7);
    return 0;
    CATCH(-1)
}

// ---- L4: src/problem.cpp + goal_types.h ----
int ref_fitness(void* h, int fk_mode, size_t n, const double* seed, const double* params, const double* base_genes, const double* genes, double* primary,
                double* secondary) {
    TRY
    Ref& r = *(Ref*)h;
    r.set_query(seed, params);
    size_t D = r.problem.active_variables.size(), T = r.problem.tip_link_indices.size();
    RobotFK fk(r.model);
    fk.initialize(r.problem.tip_link_indices);
    std::vector<Frame> null_frames(T);
    std::vector<aligned_vector<Frame>> lin;
    if (fk_mode == BIOIK_FK_LINEAR) {
        fk.applyConfiguration(full_vars(r, seed, base_genes));
        fk.initializeMutationApproximator(r.problem.active_variables);
        std::vector<const double*> ptrs(n);
        for (size_t k = 0; k < n; k++) ptrs[k] = genes + k * D;
        fk.computeApproximateMutations(n, ptrs.data(), lin);
    }
    for (size_t k = 0; k < n; k++) {
        const double* g = genes + k * D;
        if (fk_mode == BIOIK_FK_LINEAR) {
            primary[k] = r.problem.computeGoalFitness(r.problem.goals, lin[k].data(), g);
        } else {
            fk.applyConfiguration(full_vars(r, seed, g));
            primary[k] = r.problem.computeGoalFitness(r.problem.goals, fk.getTipFrames().data(), g);
        }
        secondary[k] = r.problem.computeGoalFitness(r.problem.secondary_goals, null_frames.data(), g);
    }
    return 0;
    CATCH(-1)
}
int ref_check(void* h, size_t n, const double* seed, const double* params, const double* genes, int32_t* ok) {
    TRY
    Ref& r = *(Ref*)h;
    r.set_query(seed, params);
    size_t D = r.problem.active_variables.size();
    RobotFK fk(r.model);
    fk.initialize(r.problem.tip_link_indices);
    for (size_t k = 0; k < n; k++) {
        fk.applyConfiguration(full_vars(r, seed, genes + k * D));
        ok[k] = r.problem.checkSolutionActiveVariables(fk.getTipFrames(), genes + k * D) ? 1 : 0;
    }
    return 0;
    CATCH(-1)
}

// the reference's random sources (src/ik_base.h:49-126) in the order reproduce() / step() consume them
int ref_random_probe(int seed, size_t n_gauss, double* gauss, size_t n_index, uint64_t* index16, size_t n_fast, double* fast, size_t n_rng, double* rng_uniform) {
    Random r((std::minstd_rand::result_type)seed);
    const double* g = r.fast_random_gauss_n(n_gauss);
    for (size_t i = 0; i < n_gauss; i++) gauss[i] = g[i];
    for (size_t i = 0; i < n_index; i++) index16[i] = r.fast_random_index(16);
    for (size_t i = 0; i < n_fast; i++) fast[i] = r.fast_random();
    for (size_t i = 0; i < n_rng; i++) rng_uniform[i] = r.random();
    return 0;
}

// ---- L3: src/ik_evolution_2.cpp through the reference's own factory ----
struct RefSolver {
    Ref* ref;
    std::unique_ptr<IKSolver> ik;
};
void* ref_solver_create(void* h, const double* seed, const double* params) {
    TRY
    Ref& r = *(Ref*)h;
    r.set_query(seed, params);
    auto* s = new RefSolver();
    s->ref = &r;
    s->ik.reset(IKFactory::create(r.ikparams.solver_class_name, r.ikparams));  // src/utils.h:398-444, ik_evolution_2.cpp:652-654
    s->ik->canceled = false;  // IKParallel::solve does this before every run (src/ik_parallel.h:211-212); IKBase leaves it uninitialised
    s->ik->thread_index = 0;  // IKParallel numbers its solver clones (src/ik_parallel.h:127); IKBase leaves it uninitialised
    s->ik->initialize(r.problem);
    return s;
    CATCH(nullptr)
}
void ref_solver_destroy(void* s) { delete (RefSolver*)s; }
int ref_solver_step(void* s) {
    TRY((RefSolver*)s)->ik->step();
    return 0;
    CATCH(-1)
}
extern "C" int ref_evolution_state(void* ikbase, double* genes, double* fitness, double* solution_fitness);
// population of the reference solver: species_genes [2][2][2][D] (species, individual, {genes,gradients}, gene), fitness [2]
int ref_solver_state(void* sp, double* species_genes, double* species_fitness, double* solution_fitness) {
    return ref_evolution_state(((RefSolver*)sp)->ik.get(), species_genes, species_fitness, solution_fitness);
}
// n queries through ONE solver object, as the plugin does (kinematics_plugin.cpp:273 creates IKParallel once, :566-578
// re-initialises it per query): per query the island loop of src/ik_parallel.h:160-184 in budget form — step(); exact FK of
// getSolution(); checkSolution; computeFitness — until success or max_steps.  This is the reference's own CPU path.
int ref_solve_batch(void* h, size_t n, const double* seeds, const double* params, int max_steps, double* solutions, double* fitness, int32_t* success,
                    int32_t* steps) {
    TRY
    Ref& r = *(Ref*)h;
    size_t V = r.model->getVariableCount(), P = (size_t)r.P;
    if (!r.batch_solver) r.batch_solver.reset(IKFactory::create(r.ikparams.solver_class_name, r.ikparams));  // once, like the plugin
    IKSolver* ik = r.batch_solver.get();
    ik->thread_index = 0;
    std::vector<double> zero(1, 0.0);
    for (size_t q = 0; q < n; q++) {
        r.set_query(seeds + q * V, P ? params + q * P : zero.data());
        ik->canceled = false;
        ik->initialize(r.problem);
        int st = 0;
        bool ok = false;
        double fit = DBL_MAX;
        std::vector<double> sol(seeds + q * V, seeds + (q + 1) * V);
        while (st < max_steps) {
            ik->step();
            st++;
            sol = ik->getSolution();
            ik->model.applyConfiguration(sol);
            ok = ik->checkSolution(sol, ik->model.getTipFrames());
            fit = ik->computeFitness(sol, ik->model.getTipFrames());
            if (ok) break;
        }
        for (size_t v = 0; v < V; v++) solutions[q * V + v] = sol[v];
        fitness[q] = fit, success[q] = ok ? 1 : 0, steps[q] = st;
    }
    return 0;
    CATCH(-1)
}

// The same n queries under the reference's WALL-CLOCK loop (src/ik_parallel.h:160-184, one solver thread): steps in groups of four while the clock is below the
// call's timeout, the success test behind every group, at least one group -- what searchPositionIK does with its `timeout` argument (kinematics_plugin.cpp:504,
// 566-578).  seconds [n]: the wall time of every query from the moment its goals are set (the plugin's t0) to the end of its loop.  bench.py's `one_pose_timeouts`.
int ref_solve_batch_timeout(void* h, size_t n, const double* seeds, const double* params, double timeout, double* solutions, double* fitness, int32_t* success,
                            int32_t* steps, double* seconds) {
    TRY
    Ref& r = *(Ref*)h;
    size_t V = r.model->getVariableCount(), P = (size_t)r.P;
    if (!r.batch_solver) r.batch_solver.reset(IKFactory::create(r.ikparams.solver_class_name, r.ikparams));  // once, like the plugin
    IKSolver* ik = r.batch_solver.get();
    ik->thread_index = 0;
    std::vector<double> zero(1, 0.0);
    typedef std::chrono::steady_clock clock;
    for (size_t q = 0; q < n; q++) {
        const clock::time_point t0 = clock::now();
        const clock::time_point deadline = t0 + std::chrono::duration_cast<clock::duration>(std::chrono::duration<double>(timeout));
        r.set_query(seeds + q * V, P ? params + q * P : zero.data());
        ik->canceled = false;
        ik->initialize(r.problem);
        int st = 0;
        bool ok = false;
        double fit = DBL_MAX;
        std::vector<double> sol(seeds + q * V, seeds + (q + 1) * V);
        for (size_t iteration = 0; clock::now() < deadline || iteration == 0; iteration++) {
            ik->step();
            st++;
            for (int it2 = 1; it2 < 4; it2++)
                if (clock::now() < deadline) ik->step(), st++;
            sol = ik->getSolution();
            ik->model.applyConfiguration(sol);
            ok = ik->checkSolution(sol, ik->model.getTipFrames());
            fit = ik->computeFitness(sol, ik->model.getTipFrames());
            if (ok) break;
        }
        seconds[q] = std::chrono::duration<double>(clock::now() - t0).count();
        for (size_t v = 0; v < V; v++) solutions[q * V + v] = sol[v];
        fitness[q] = fit, success[q] = ok ? 1 : 0, steps[q] = st;
    }
    return 0;
    CATCH(-1)
}

// solution [V]; exact-FK fitness and success of the solution as src/ik_parallel.h:173-181 computes them
int ref_solver_result(void* sp, double* solution, double* fitness, int32_t* success) {
    TRY
    RefSolver& s = *(RefSolver*)sp;
    const std::vector<double>& sol = s.ik->getSolution();
    for (size_t i = 0; i < sol.size(); i++) solution[i] = sol[i];
    s.ik->model.applyConfiguration(sol);
    *success = s.ik->checkSolution(sol, s.ik->model.getTipFrames()) ? 1 : 0;
    *fitness = s.ik->computeFitness(sol, s.ik->model.getTipFrames());
    return 0;
    CATCH(-1)
}
}
